import sys, struct, zlib, json, subprocess, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/cpp-fluid-particles_amd')
from oracle import oracle as O
import sphx
B = '/root/repo/tools/ref_oracle/_build/'
P, fluid, shell = O.scene(12)
box = sphx.sample_box((0.40, 0.0, 0.10), (0.46, 0.15, 0.40), 0.02)
ball = sphx.sample_sphere((0.25, 0.02, 0.25), 0.018, 0.01)
ramp = sphx.sample_triangles(np.float32([[0.05, 0.0, 0.05, 0.13, 0.0, 0.05, 0.05, 0.06, 0.45], [0.13, 0.0, 0.05, 0.13, 0.06, 0.45, 0.05, 0.06, 0.45]]), 0.02)
boundary = np.concatenate([shell, box, ball, ramp]).astype(np.float32)
vel = np.zeros_like(fluid); vel[:, 1] = -1.5; vel[:, 0] = 0.8
D = B + 'dumps/obst'; os.makedirs(D, exist_ok=True)
with open(D + '/in.bin', 'wb') as f:
    f.write(struct.pack('fii', float(P.space[0]), len(fluid), len(boundary))); f.write(fluid.tobytes()); f.write(vel.tobytes()); f.write(boundary.tobytes())
out = {"_scene": "nx = 12 dam-break block with velocity (0.8, -1.5, 0), boundary = shell + sphx_sample_box((0.40,0,0.10),(0.46,0.15,0.40),0.02) + sphx_sample_sphere((0.25,0.02,0.25),0.018,0.01) + sphx_sample_triangles(two-triangle ramp, 0.02), concatenated in that order"}
for sid, name, dt in ((1, "dfsph", 0.002), (0, "wcsph", 0.001)):
    steps = 20
    subprocess.check_call([B + 'refAf', str(sid), str(steps), str(dt), D, '10', D + '/in.bin'], stdout=open(D + '/log%d.txt' % sid, 'w'))
    raw = open(D + '/scene.bin', 'rb').read()
    n, nb = np.frombuffer(raw[:8], np.int32)
    fl = np.frombuffer(raw[8:8 + 12 * n], np.float32).reshape(n, 3)
    o0 = np.lexsort((fluid[:, 2], fluid[:, 1], fluid[:, 0])); o1 = np.lexsort((fl[:, 2], fl[:, 1], fl[:, 0]))
    orig = np.empty(n, np.int64); orig[o1] = o0
    assert np.array_equal(fluid[orig], fl)
    bm = np.fromfile(D + '/bmass.bin', np.float32)[:nb]
    P.solver = sid; P.dt = dt
    if sid == 0: P.pow7_mode = 1
    s = O.System(P, fluid, boundary, ctor_step=False)
    print(name, "boundary masses bit-identical:", np.array_equal(bm.view(np.uint32), s.get(O.F_BMASS).view(np.uint32)), len(bm))
    ids = s.get(O.F_ID); s.set(O.F_VEL, vel[ids]); s.step()
    rows = []
    for step in range(0, steps + 1):
        if step: s.step()
        if step % 10: continue
        a = np.fromfile('%s/s%d_%04d.bin' % (D, sid, step), np.float32)
        p = np.empty((n, 3), np.float32); v = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
        p[orig] = a[:3 * n].reshape(n, 3); v[orig] = a[3 * n:6 * n].reshape(n, 3); r[orig] = a[6 * n:]
        ids = s.get(O.F_ID)
        po = np.empty_like(p); vo = np.empty_like(v); ro = np.empty_like(r)
        po[ids] = s.get(O.F_POS); vo[ids] = s.get(O.F_VEL); ro[ids] = s.get(O.F_DENSITY)
        print(name, step, [np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in ((p, po), (v, vo), (r, ro))], s.iters(), flush=True)
        rows.append({"step": step, "rho_mean": float("%.9g" % r.mean(dtype=np.float64)), "crc32_pos": zlib.crc32(p.tobytes()), "crc32_vel": zlib.crc32(v.tobytes()),
                     "crc32_density": zlib.crc32(r.tobytes()), **({"iters_div_den": list(s.iters())} if sid == 1 else {})})
    out[name] = {"dt": dt, "states": rows}
    out["crc32_boundary_mass_sorted"] = zlib.crc32(bm.tobytes()); out["boundary_count"] = int(nb)
json.dump(out, open(B + 'obst_anchors.json', 'w'), indent=1)
