"""ctypes binding of the CPU oracle (oracle/sph_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (cpp-fluid-particles_amd/) never does and has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsph_oracle.so")

WCSPH, DFSPH, PBD = 0, 1, 2

# field ids (same numbering in include/sphx_c.h)
(F_POS, F_VEL, F_DENSITY, F_PRESSURE, F_MASS, F_CELL, F_CELLSTART_F, F_CELLSTART_B, F_ID, F_BPOS,
 F_BMASS, F_ALPHA, F_KAPPA, F_ERROR, F_WARM, F_POS_LAST, F_LAMBDA, F_BUF3) = range(18)


class Params(C.Structure):
    """Field-for-field mirror of sphx_params (include/sphx_c.h) and oracle_params."""
    _fields_ = [
        ("space", C.c_float * 3), ("cells", C.c_int * 3),
        ("cell_length", C.c_float), ("radius", C.c_float), ("dt", C.c_float), ("m0", C.c_float),
        ("rho0", C.c_float), ("rho_boundary", C.c_float), ("stiff", C.c_float), ("visc", C.c_float),
        ("surface_tension", C.c_float), ("air_pressure", C.c_float), ("gravity", C.c_float * 3),
        ("solver", C.c_int),
        ("dfsph_density_thr", C.c_float), ("dfsph_divergence_thr", C.c_float),
        ("dfsph_max_iter", C.c_int), ("dfsph_fixed_div", C.c_int), ("dfsph_fixed_den", C.c_int),
        ("pbd_iters", C.c_int), ("pbd_xsph_c", C.c_float), ("pbd_relaxation", C.c_float),
        ("pow7_mode", C.c_int), ("xsph_mode", C.c_int), ("reserved", C.c_int * 4),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "sph_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsph_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.oracle_step.restype = C.c_float
        L.oracle_step.argtypes = [C.c_void_p]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_get.restype = C.c_longlong
        L.oracle_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        L.oracle_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        L.oracle_iters.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_scene_params.argtypes = [C.c_int, C.POINTER(Params)]
        L.oracle_scene_counts.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_scene_fill.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_eval_kernels.argtypes = [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_w_promote.argtypes = [C.c_int]
        L.oracle_run_phase.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_count.argtypes = [C.c_void_p, C.c_int]
        L.oracle_error_total_fixed.restype = C.c_longlong
        L.oracle_error_total_fixed.argtypes = [C.c_void_p, C.c_int, C.c_int]
        assert L.oracle_sizeof_params() == C.sizeof(Params)
        _lib = L
    return _lib


def set_w_promote(on):
    """Diagnostic arithmetic of a g++ host compile of the reference text (see sph_oracle.c, g_w_promote)."""
    lib().oracle_set_w_promote(int(bool(on)))


def scene(nx):
    """Dam-break scene of BASELINE.md §4 (nx=24 is the reference's main.cpp scene)."""
    L = lib()
    P = Params()
    L.oracle_scene_params(nx, C.byref(P))
    n, nb = C.c_int(), C.c_int()
    L.oracle_scene_counts(nx, C.byref(n), C.byref(nb))
    fluid = np.empty((n.value, 3), np.float32)
    boundary = np.empty((nb.value, 3), np.float32)
    L.oracle_scene_fill(nx, fluid.ctypes.data, boundary.ctypes.data)
    return P, fluid, boundary


_SHAPES = {F_POS: (3, np.float32), F_VEL: (3, np.float32), F_BPOS: (3, np.float32),
           F_POS_LAST: (3, np.float32), F_BUF3: (3, np.float32), F_CELL: (1, np.int32),
           F_CELLSTART_F: (1, np.int32), F_CELLSTART_B: (1, np.int32), F_ID: (1, np.int32)}


class System:
    def __init__(self, params, fluid, boundary, ctor_step=True, threads=0):
        L = lib()
        if threads:
            L.oracle_set_threads(threads)
        fluid = np.ascontiguousarray(fluid, np.float32)
        boundary = np.ascontiguousarray(boundary, np.float32)
        self.n, self.nb = len(fluid), len(boundary)
        self.C = params.cells[0] * params.cells[1] * params.cells[2]
        self.params = params
        self._h = L.oracle_create(C.byref(params), fluid.ctypes.data, self.n,
                                  boundary.ctypes.data, self.nb, int(ctor_step))

    def step(self):
        return lib().oracle_step(self._h)

    def _count(self, field):
        if field in (F_CELLSTART_F, F_CELLSTART_B):
            return self.C + 1
        if field in (F_BPOS, F_BMASS):
            return self.nb
        return self.n

    def get(self, field):
        comps, dt = _SHAPES.get(field, (1, np.float32))
        out = np.empty((self._count(field), comps) if comps > 1 else (self._count(field),), dt)
        got = lib().oracle_get(self._h, field, out.ctypes.data, out.nbytes)
        assert got == out.nbytes, (field, got, out.nbytes)
        return out

    def set(self, field, arr):
        arr = np.ascontiguousarray(arr)
        assert lib().oracle_set(self._h, field, arr.ctypes.data, arr.nbytes) == 0

    def set_count(self, n):
        assert lib().oracle_set_count(self._h, n) == 0
        self.n = n

    def run_phase(self, phase):
        assert lib().oracle_run_phase(self._h, phase) == 0

    def error_total_fixed(self, lo, hi):
        return int(lib().oracle_error_total_fixed(self._h, lo, hi))

    def iters(self):
        a, b = C.c_int(), C.c_int()
        lib().oracle_iters(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def eval_kernels(r3, R):
    r3 = np.ascontiguousarray(r3, np.float32)
    n = len(r3)
    W = np.empty(n, np.float32); G = np.empty((n, 3), np.float32)
    V = np.empty(n, np.float32); S = np.empty((n, 3), np.float32)
    lib().oracle_eval_kernels(r3.ctypes.data, n, R, W.ctypes.data, G.ctypes.data, V.ctypes.data, S.ctypes.data)
    return W, G, V, S
