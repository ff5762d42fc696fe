import sys, struct, zlib, json, subprocess, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/cpp-fluid-particles_amd')
from oracle import oracle as O
B = '/root/repo/tools/ref_oracle/_build/'
def splash(n, P, seed):
    rng = np.random.default_rng(seed)
    lo = 0.02 * P.space[0]
    pos = rng.uniform(lo, 0.5 * P.space[0], (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(lo, 0.35 * P.space[1], n).astype(np.float32)
    vel = rng.normal(0, 0.8, (n, 3)).astype(np.float32)
    return pos, vel
out = {}
nx = 12
for sid, name, dt in ((0, "wcsph", 0.001), (1, "dfsph", 0.001), (2, "pbd", 0.001)):
    P, fluid, boundary = O.scene(nx)
    pos, vel = splash(len(fluid), P, 5 + sid)
    D = B + 'dumps/%s' % name; os.makedirs(D, exist_ok=True)
    with open(D + '/in.bin', 'wb') as f:
        f.write(struct.pack('fii', float(P.space[0]), len(pos), len(boundary))); f.write(pos.tobytes()); f.write(vel.tobytes()); f.write(boundary.tobytes())
    steps = 30
    subprocess.check_call([B + 'refAf', str(sid), str(steps), str(dt), D, '10', D + '/in.bin'], stdout=open(D + '/log.txt', 'w'))
    raw = open(D + '/scene.bin', 'rb').read()
    n, nb = np.frombuffer(raw[:8], np.int32)
    fl = np.frombuffer(raw[8:8 + 12 * n], np.float32).reshape(n, 3)      # presorted order = id order of the dumps
    # presorted index -> original index (positions are unique)
    o0 = np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0])); o1 = np.lexsort((fl[:, 2], fl[:, 1], fl[:, 0]))
    orig = np.empty(n, np.int64); orig[o1] = o0
    assert np.array_equal(pos[orig], fl)
    # oracle, the reference flow: searches, velocities in sorted order, then the constructor's step
    P.solver = sid; P.dt = dt; P.pbd_iters = 20
    if sid == 0: P.pow7_mode = 1
    if sid == 2: P.xsph_mode = 1
    s = O.System(P, pos, boundary, ctor_step=False)
    ids = s.get(O.F_ID); s.set(O.F_VEL, vel[ids])
    s.step()
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
        same = [np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in ((p, po), (v, vo), (r, ro))]
        print(name, step, same, s.iters(), flush=True)
        rows.append({"step": step, "rho_mean": float("%.9g" % r.mean(dtype=np.float64)), "rho_max": float("%.9g" % r.max()),
                     "crc32_pos": zlib.crc32(p.tobytes()), "crc32_vel": zlib.crc32(v.tobytes()), "crc32_density": zlib.crc32(r.tobytes())})
    out[name] = {"dt": dt, "seed": 5 + sid, "states": rows}
json.dump(out, open(B + 'splash_anchors.json', 'w'), indent=1)
