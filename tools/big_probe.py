"""large single-GPU scenes: creation / step time and row statistics (python tools/big_probe.py nx [flags])"""
import sys, time, numpy as np
sys.path[:0] = ["cpp-fluid-particles_amd"]
import torch, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
for nx in [int(a) for a in sys.argv[1].split(",")]:
    for flags in [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]:
        P, fluid, boundary = sphx.scene(nx)
        P.solver = sphx.WCSPH; P.dt = 0.001; P.reserved[0] = flags
        free0 = torch.cuda.mem_get_info()[0]
        t = time.time(); s = sphx.System(P, fluid, boundary); tc = time.time() - t
        t = time.time(); s.step(); ts = time.time() - t
        tot, mx, h = s.row_stats()
        used = free0 - torch.cuda.mem_get_info()[0]
        d = s.get(sphx.F_DENSITY)
        print("nx %d flags %d: n %d create %.3f s step %.1f ms rows: pairs %d longest %d density %.4f..%.4f device memory %.2f GB = %.0f B/particle" % (nx, flags, len(fluid), tc, ts * 1e3, tot, mx, d.min(), d.max(), used / 1e9, used / len(fluid)), flush=True)
        s.close()
