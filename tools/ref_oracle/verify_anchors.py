"""Re-runs the diagnostic build (make) on the dam-break scene and checks that it reproduces the CRCs recorded in
tests/golden/refsrc_anchors.json (variant float_fabs; DFSPH to step 30 by default -- `all` runs every recorded state of the
three solvers, a few minutes).  Needs /root/reference (build container only)."""
import json, os, subprocess, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O
subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)
B = os.path.join(HERE, "_build")
A = json.load(open(os.path.join(HERE, "..", "..", "tests", "golden", "refsrc_anchors.json")))["variants"]["float_fabs"]
full = len(sys.argv) > 1 and sys.argv[1] == "all"
P, f0, b0 = O.scene(24)
bad = 0
for sid, name, every in ((1, "dfsph", 10), (0, "wcsph", 50), (2, "pbd", 20)):
    states = [s for s in A[name]["states"] if full or (name == "dfsph" and s["step"] <= 30)]
    if not states:
        continue
    D = os.path.join(B, "verify_" + name); os.makedirs(D, exist_ok=True)
    subprocess.check_call([os.path.join(B, "refAf"), str(sid), str(states[-1]["step"]), str(A[name]["dt"]), D, str(every)], stdout=open(os.path.join(D, "log.txt"), "w"))
    raw = open(os.path.join(D, "scene.bin"), "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    fl = np.frombuffer(raw[8:8 + 12 * n], np.float32).reshape(n, 3)
    o0 = np.lexsort((f0[:, 2], f0[:, 1], f0[:, 0])); o1 = np.lexsort((fl[:, 2], fl[:, 1], fl[:, 0]))
    orig = np.empty(n, np.int64); orig[o1] = o0          # presorted index -> index in main.cpp's fill order
    for st in states:
        a = np.fromfile(os.path.join(D, "s%d_%04d.bin" % (sid, st["step"])), np.float32)
        p = np.empty((n, 3), np.float32); v = np.empty((n, 3), np.float32); r = np.empty(n, np.float32)
        p[orig] = a[:3 * n].reshape(n, 3); v[orig] = a[3 * n:6 * n].reshape(n, 3); r[orig] = a[6 * n:]
        ok = (zlib.crc32(p.tobytes()), zlib.crc32(v.tobytes()), zlib.crc32(r.tobytes())) == (st["crc32_pos"], st["crc32_vel"], st["crc32_density"])
        print(name, "step", st["step"], "reproduced" if ok else "DIFFERS")
        bad += not ok
sys.exit(1 if bad else 0)
