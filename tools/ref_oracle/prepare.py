"""Copies the reference's solver sources from /root/reference/src into tools/ref_oracle/_build/src (git-ignored, never shipped) with
the two mechanical edits the stand-in-header CPU build needs: kernel launches `k <<<grid, block>>> (args` become
`LAUNCH(grid, block, k, args` (a serial loop, shim/cuda_runtime.h), and DFSPHSolver::step publishes the iteration counts it
computes and drops (DFSPHSolver.cu:49,65).  Nothing of the result is committed."""
import os, re, sys
SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "src")
os.makedirs(OUT, exist_ok=True)
launch = re.compile(r"(\b\w+)\s*<<\s*<\s*(.*?)\s*>>\s*>\s*\(", re.S)
for name in sorted(os.listdir(SRC)):
    if not name.endswith((".cu", ".cuh", ".h")) or name in ("vbo.cu", "ShaderUtility.h"):
        continue
    text = open(os.path.join(SRC, name), encoding="utf-8", errors="ignore").read()
    if name.endswith(".cu"):
        text = launch.sub(lambda m: "LAUNCH(%s, %s, " % (m.group(2), m.group(1)), text)
    if name == "DFSPHSolver.cu":
        text = text.replace("auto it_div = correctDivergenceError(", "extern int g_it_div; g_it_div = correctDivergenceError(")
        text = text.replace("auto it_den = project(", "extern int g_it_den; g_it_den = project(")
    open(os.path.join(OUT, name), "w").write(text)
print("prepared", OUT)
