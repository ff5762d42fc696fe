import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
D = '/root/repo/tools/ref_oracle/_build/' + __import__('os').environ.get('DUMP','dump')
solver = int(sys.argv[1]); steps = int(sys.argv[2]); dt = float(sys.argv[3]); every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
raw = open(D + '/scene.bin', 'rb').read()
n, nb = np.frombuffer(raw[:8], np.int32)
fl = np.frombuffer(raw[8:8 + 12 * n], np.float32).reshape(n, 3)
bd = np.frombuffer(raw[8 + 12 * n:], np.float32).reshape(nb, 3)
P, f0, b0 = O.scene(24)
print('boundary equal', np.array_equal(bd, b0), 'fluid set equal', np.array_equal(np.sort(fl.view('V12').ravel()), np.sort(f0.view('V12').ravel())))
P.solver = solver; P.dt = dt
for k, v in [a.split('=') for a in sys.argv[5:]]:
    setattr(P, k, type(getattr(P, k))(float(v)) if not isinstance(getattr(P, k), int) else int(v))
O.set_w_promote(int(__import__('os').environ.get('WP','0')))
s = O.System(P, fl, bd)
def cmp(step):
    a = np.fromfile('%s/s%d_%04d.bin' % (D, solver, step), np.float32)
    pa = a[:3 * n].reshape(n, 3); va = a[3 * n:6 * n].reshape(n, 3); da = a[6 * n:]
    ids = s.get(O.F_ID)
    pb = np.empty_like(pa); vb = np.empty_like(va); db = np.empty_like(da)
    pb[ids] = s.get(O.F_POS); vb[ids] = s.get(O.F_VEL); db[ids] = s.get(O.F_DENSITY)
    r = []
    for nm, x, y in (('pos', pa, pb), ('vel', va, vb), ('rho', da, db)):
        ne = (x.view(np.uint32) != y.view(np.uint32)); r.append('%s diff %d max %.3e' % (nm, ne.sum(), np.abs(x - y).max()))
    print(step, s.iters(), ' | '.join(r), 'rho_mean %.6f' % db.mean(dtype=np.float64)); sys.stdout.flush()
cmp(0)
for st in range(1, steps + 1):
    s.step()
    if st % every == 0: cmp(st)
