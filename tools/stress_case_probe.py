"""One case of tools/stress_parity.py's tolerance mode under the magnifying glass (GPU box): the engine's deviation from the oracle
for every quad mask, next to the oracle's OWN sensitivity on that state (the same run from positions moved by one ulp).
python tools/stress_case_probe.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
from oracle import oracle as O
import stress_parity as S


def setup(seed):
    rng = np.random.default_rng(seed)
    nx = int(rng.choice([8, 10, 12, 16]))
    P, fluid, boundary = sphx.scene(nx)
    solver = int(rng.integers(0, 3))
    P.solver = solver; P.dt = float(rng.choice([0.0005, 0.001])); P.pbd_iters = int(rng.integers(1, 4))
    P.dfsph_fixed_div, P.dfsph_fixed_den = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    n = int(rng.integers(64, len(fluid)))
    pos, vel = S.make_state(rng, n, P)
    vel *= np.float32(0.3)
    return P, boundary, pos, vel


def oracle_run(P, boundary, pos, vel, steps=2):
    Po = O.Params()
    for name, _ in P._fields_:
        setattr(Po, name, getattr(P, name))
    Po.reserved[3] = 0
    o = O.System(Po, pos, boundary, ctor_step=False)
    ids = o.get(O.F_ID); o.set(O.F_VEL, vel[ids])
    for _ in range(steps): o.step()
    out = {k: o.get(getattr(O, "F_" + k)).astype(np.float64) for k in ("POS", "VEL", "DENSITY")}
    out["ID"] = o.get(O.F_ID)
    o.close()
    return out


def main():
    seed = int(sys.argv[1])
    P, boundary, pos, vel = setup(seed)
    ref = oracle_run(P, boundary, pos, vel)
    scale = {"POS": P.space[0], "DENSITY": max(float(np.abs(ref["DENSITY"]).max()), 1e-6), "VEL": max(float(np.abs(ref["VEL"]).max()), 1e-3)}
    print("seed", seed, "solver", P.solver, "n", len(pos), "scales", scale)
    rng = np.random.default_rng(1)
    for trial in range(3):
        p2 = np.nextafter(pos, np.where(rng.random(pos.shape) < 0.5, -np.inf, np.inf).astype(np.float32)).astype(np.float32)
        pert = oracle_run(P, boundary, p2, vel)
        if not np.array_equal(pert["ID"], ref["ID"]):
            print("  oracle, positions +-1 ulp: particle order differs"); continue
        print("  oracle, positions +-1 ulp:", {k: "%.2e" % float(np.abs(pert[k] - ref[k]).max() / scale[k]) for k in scale})
    for mask in ("0", "1", "7", "15", "255"):
        os.environ["SPHX_QUAD_MASK_TOL"] = mask
        P.reserved[3] = 1
        g = sphx.System(P, pos, boundary, ctor_step=False)
        ids = g.get(sphx.F_ID); g.set(sphx.F_VEL, vel[ids])
        g.step(); g.step()
        same = np.array_equal(g.get(sphx.F_ID), ref["ID"])
        dev = {k: "%.2e" % float(np.abs(g.get(getattr(sphx, "F_" + k)).astype(np.float64) - ref[k]).max() / scale[k]) for k in scale} if same else "order differs"
        print("  engine tolerance, quad mask", mask, ":", dev)
        g.close()
        P.reserved[3] = 0


if __name__ == "__main__":
    main()
