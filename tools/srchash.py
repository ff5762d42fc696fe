"""Hash of everything that determines the engine's generated code (csrc/, include/, the Makefile).
bench.py only trusts a committed profiles/traffic.json entry whose recorded hash equals the hash of
the sources the running libsphx.so was built from; tools/profile_summarize.py records it."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def engine_source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "cpp-fluid-particles_amd", "csrc", "*")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")) +
                   [os.path.join(ROOT, "cpp-fluid-particles_amd", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(engine_source_hash())
