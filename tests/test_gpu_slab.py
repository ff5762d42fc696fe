"""GPU tests of the x-slab driver with the HIP engine: ranks share the single GPU of the test box and
talk over gloo (staged through the host); the result must equal the single-domain ORACLE result
bit for bit.  (RCCL transport itself needs a multi-GPU node; the driver code is the same.)"""
import numpy as np
import pytest

from conftest import assert_bit_equal
import slab_worker
from test_slab_cpu import _free_port, _single_domain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,solver,adaptive", [(1, "dfsph", False), (2, "dfsph", False), (3, "dfsph", False),
                                                   (2, "wcsph", False), (2, "dfsph", True), (2, "pbd", False),
                                                   (3, "pbd", False)])
def test_hip_slab_driver_matches_single_domain_oracle(oracle, tmp_path, world, solver, adaptive):
    import torch.multiprocessing as mp
    nx, steps, seed = 12, 6, 17
    mp.spawn(slab_worker.run, args=(world, _free_port(), "gloo", "hip", nx, steps, str(tmp_path), seed, solver, adaptive),
             nprocs=world, join=True)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    ids = np.concatenate([p["ids"] for p in parts])
    n = len(ids)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.int32))
    order = np.argsort(ids)
    rp, rv, rd, it = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    if adaptive:
        assert all(tuple(p["iters"]) == it for p in parts)
    assert_bit_equal(np.concatenate([p["pos"] for p in parts])[order], rp, "slab pos")
    assert_bit_equal(np.concatenate([p["vel"] for p in parts])[order], rv, "slab vel")
    assert_bit_equal(np.concatenate([p["density"] for p in parts])[order], rd, "slab density")
    if world > 1:
        assert sum(int(p["migrated"]) for p in parts) > 0
