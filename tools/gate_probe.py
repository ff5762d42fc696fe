"""What the gated launches of a terminated adaptive DFSPH loop cost on the reference scene: step time in free fall (1 divergence + 2 density
iterations) against the number of iterations enqueued (dfsph_max_iter), with the loop tail (one persistent launch) and without, graph replay
and eager.   python tools/gate_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
for graph in (1, 0):
    for tail in (1, 0):
        for mi in (20, 6, 3):
            if tail: os.environ.pop("SPHX_DFSPH_NO_TAIL", None)
            else: os.environ["SPHX_DFSPH_NO_TAIL"] = "1"
            if graph: os.environ.pop("SPHX_NO_GRAPH", None)
            else: os.environ["SPHX_NO_GRAPH"] = "1"
            P, f, b = sphx.scene(24)
            P.solver = sphx.DFSPH; P.dfsph_max_iter = mi
            s = sphx.System(P, f, b)
            s.step(); s.step_n(5)
            out = ["%.3f" % (s.step_n(20) / 20) for rep in range(2)]
            s.step()
            print("graph %d tail %d max_iter %2d  ms/step %s  iterations %s" % (graph, tail, mi, out, s.iters()), flush=True)
            s.close()
