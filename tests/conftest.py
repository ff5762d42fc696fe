import os
import sys

import numpy as np
import pytest

# The oracle's OpenMP regions are tiny in these tests; on a many-core box (or a CPU-limited container
# that still reports every host core) a full-width team per region costs far more than the work.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
ORACLE_TEST_THREADS = 16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sphx():
    import sphx as mod
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    # the tests select engine variants with SPHX_* environment variables (monkeypatch.setenv); the library reads none itself:
    # tests/tuning_env.py turns them into the sphx_tuning block before every system creation / step
    import tuning_env
    return tuning_env.install(mod)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as mod
    L = mod.lib()
    L.oracle_set_threads(min(L.oracle_max_threads(), ORACLE_TEST_THREADS))
    return mod


def same_params(dst, src):
    """copy every field of a Params-like ctypes struct into another (oracle <-> sphx)."""
    for name, _ in src._fields_:
        setattr(dst, name, getattr(src, name))
    return dst


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(got, want, what):
    got = np.ascontiguousarray(got); want = np.ascontiguousarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    gb, wb = bits(got), bits(want)
    if not np.array_equal(gb, wb):
        bad = np.flatnonzero(gb.reshape(-1) != wb.reshape(-1))
        i = bad[0]
        raise AssertionError("%s: %d of %d elements differ bitwise; first at flat %d: got %r want %r" % (
            what, bad.size, gb.size, i, got.reshape(-1)[i], want.reshape(-1)[i]))
