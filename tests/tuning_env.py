"""Test / tool infrastructure: SPHX_* environment variables -> sphx_tuning (include/sphx_c.h).

Until r04 the engine library read ~27 SPHX_* environment variables itself.  It reads none now: behaviour switches are fields of
the process-wide `sphx_tuning` block installed through the C ABI (sphx_set_tuning).  The test suite and the measurement tools keep
their habit of selecting engine variants with environment variables (monkeypatch.setenv(...) in a test, VAR=... on a tool's command
line): install(sphx) wraps the binding's entry points so that, whenever a system or slab group is created or stepped, the variables
present in os.environ are translated into a tuning block and installed first.  The product (cpp-fluid-particles_amd/, libsphx.so)
knows nothing of this file."""
import os

_PRESENT = lambda v: 1                      # the variable's presence is the switch
_INT = int


def _comm_priority(v):
    return {"0": 1, "default": 1, "low": 2}.get(v, 0)


# variable -> (sphx_tuning field, conversion)
ENV = {
    "SPHX_ENGINE_FLAGS": ("engine_flags", _INT),
    "SPHX_NBR_CAP": ("row_capacity", _INT),
    "SPHX_QUAD_MASK": ("quad_mask", _INT),
    "SPHX_DUO_MASK": ("duo_mask", _INT),
    "SPHX_QUAD_MASK_TOL": ("quad_mask_tol", _INT),
    "SPHX_TOL_STRICT_RATE": ("tol_strict_rate", lambda v: int(int(v) != 0)),
    "SPHX_BRICK": ("brick", lambda v: int(int(v) != 0)),
    "SPHX_BRICK_MIN": ("brick_min", _INT),
    "SPHX_RANGE_ORDER": ("range_order", lambda v: int(int(v) != 0)),
    "SPHX_RANGE_ORDER_MIN": ("range_order_min", _INT),
    "SPHX_FORCE_TILE_ORDER": ("force_tile_order", _PRESENT),
    "SPHX_NO_FASTMATH": ("no_fastmath", _PRESENT),
    "SPHX_NO_GRAPH": ("no_graph", _PRESENT),
    "SPHX_GRAPH_DEBUG": ("graph_debug", _PRESENT),
    "SPHX_DFSPH_HOST_LOOP": ("dfsph_host_loop", _PRESENT),
    "SPHX_DFSPH_WINDOW": ("dfsph_window", lambda v: max(0, int(v))),
    "SPHX_DFSPH_NO_TAIL": ("dfsph_no_tail", _PRESENT),
    "SPHX_DFSPH_TAIL_FLAT": ("dfsph_tail_flat", _PRESENT),
    "SPHX_GROUP_BUILD_MAX": ("group_build_max", _INT),
    "SPHX_NO_KICK_FUSION": ("no_kick_fusion", _PRESENT),
    "SPHX_PBD_SKIN": ("pbd_skin", float),
    "SPHX_PBD_SKIN_FIXED": ("pbd_skin_fixed", _PRESENT),
    "SPHX_PBD_NO_PARTIAL": ("pbd_no_partial", _PRESENT),
    "SPHX_PERSIST_CONTROLLER": ("persist_controller", lambda v: int(int(v) != 0)),
    "SPHX_SLAB_EDGE_STREAM": ("slab_edge_stream", lambda v: int(v != "0")),
    "SPHX_COMM_PRIORITY": ("slab_comm_priority", _comm_priority),
}


def from_environment(sphx, environ=None):
    environ = os.environ if environ is None else environ
    t = sphx.default_tuning()
    for var, (field, conv) in ENV.items():
        if var in environ:
            setattr(t, field, conv(environ[var]))
    return t


def apply(sphx):
    sphx.set_tuning(from_environment(sphx))


def install(sphx):
    """wrap the creation / stepping entry points of the binding module so that the environment is translated first (idempotent)"""
    if getattr(sphx, "_tuning_env_installed", False):
        return sphx
    sphx._tuning_env_installed = True

    def wrap(owner, name):
        inner = getattr(owner, name)

        def outer(*a, **k):
            apply(sphx)
            return inner(*a, **k)
        outer.__name__ = name
        outer.__doc__ = inner.__doc__
        setattr(owner, name, outer)

    for name in ("__init__", "step", "step_n", "profile_step"):
        wrap(sphx.System, name)
    for name in ("__init__", "step"):
        wrap(sphx.SlabGroup, name)
    wrap(sphx, "load_snapshot")
    wrap(sphx, "fastmath_selftest")
    return sphx
