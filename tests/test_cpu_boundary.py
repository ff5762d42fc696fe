"""CPU tests (no GPU) of the drop-in boundary: the C ABI library loads and exports every symbol the
header declares, the host-side pieces (scene generator, parameter block) agree with the oracle,
and the product refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal


def _declared(header="sphx_c.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sphx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sphx):
    L = sphx.lib()
    names = _declared()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(L, nm), "libsphx.so does not export %s declared in include/sphx_c.h" % nm
    assert sorted(sphx.EXPORTS) == names, "sphx.py EXPORTS out of date with include/sphx_c.h"
    slab = [nm for nm in _declared("sphx_slab.h") if nm.startswith("sphx_slab_") and nm != "sphx_slab_group"]
    assert sorted(sphx.SLAB_EXPORTS) == slab, "sphx.py SLAB_EXPORTS out of date with include/sphx_slab.h"
    for nm in slab:
        assert hasattr(L, nm), "libsphx.so does not export %s declared in include/sphx_slab.h" % nm
    for nm in sphx.REFERENCE_EXPORTS:       # the reference's own extern "C" symbol (vbo.cu:46-51)
        assert hasattr(L, nm), "libsphx.so does not export the reference symbol %s" % nm


def test_param_block_layouts_match(sphx, oracle):
    assert C.sizeof(sphx.Params) == C.sizeof(oracle.Params) == sphx.lib().sphx_sizeof_params()
    assert [f[0] for f in sphx.Params._fields_] == [f[0] for f in oracle.Params._fields_]


@pytest.mark.parametrize("nx", [8, 24, 40])
def test_scene_generators_agree(sphx, oracle, nx):
    """the product's scene generator (C++) and the oracle's (C) are independent restatements of
    main.cpp:54-117; they must agree bit for bit."""
    Pg, fg, bg = sphx.scene(nx)
    Po, fo, bo = oracle.scene(nx)
    assert_bit_equal(fg, fo, "fluid"); assert_bit_equal(bg, bo, "boundary")
    for name, _ in Pg._fields_:
        a, b = getattr(Pg, name), getattr(Po, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name


def test_scene_rejects_bad_sizes(sphx):
    with pytest.raises(sphx.SphxError):
        sphx.scene(7)
    with pytest.raises(sphx.SphxError):
        sphx.scene(0)


def test_no_device_means_loud_failure(sphx):
    """there is no CPU path: on a box without a GPU, creating a system must raise."""
    if sphx.device_count() > 0:
        pytest.skip("a HIP device is present")
    P, fluid, boundary = sphx.scene(8)
    with pytest.raises(sphx.SphxError) as e:
        sphx.System(P, fluid, boundary)
    assert "no HIP device" in str(e.value)
    with pytest.raises(sphx.SphxError):
        sphx.eval_kernels(np.zeros((4, 3), np.float32), 0.04)


def test_create_validates_arguments(sphx):
    P, fluid, boundary = sphx.scene(8)
    P.pow7_mode = 1
    with pytest.raises(sphx.SphxError):
        sphx.System(P, fluid, boundary)
    P.pow7_mode = 0
    P.cells[0] = 0
    with pytest.raises(sphx.SphxError):
        sphx.System(P, fluid, boundary)


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under the package or include/ may mention it."""
    pkg = os.path.join(ROOT, "cpp-fluid-particles_amd")
    for base in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            if "build" in dirpath:
                continue
            for f in files:
                if f.endswith((".hip", ".hpp", ".h", ".py", ".cpp")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "sph_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
