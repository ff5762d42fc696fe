"""CPU stand-in for the HIP slab engine, used ONLY by the tests: the oracle behind the engine interface
of tests/slab_protocol.py (copies instead of zero-copy views), so that the slab driver
itself — particle exchange, halo schedule, adaptive termination — runs under gloo without a GPU."""
import numpy as np
import torch

from slab_protocol import cell_column


class OracleSlabEngine:
    zero_copy = False

    def __init__(self, O, params, cap, boundary_pos, boundary_mass):
        self.O, self.cap = O, cap
        self.sys = O.System(params, np.zeros((cap, 3), np.float32), boundary_pos, ctor_step=False)
        if len(boundary_pos):
            self.sys.set(O.F_BMASS, np.ascontiguousarray(boundary_mass, np.float32))
        self.C = self.sys.C
        self.map = {"pos": O.F_POS, "vel": O.F_VEL, "ids": O.F_ID, "vel_nbr": O.F_VEL, "cg_nbr": O.F_BUF3,
                    "density": O.F_DENSITY, "pressure": O.F_PRESSURE}
        if params.solver == O.DFSPH:
            self.map["warm"] = O.F_WARM
            self.map["kappa"] = O.F_KAPPA
        if params.solver == O.PBD:
            self.map["pos_last"] = O.F_POS_LAST
            self.map["lambda"] = O.F_LAMBDA
            self.map["pos_nbr"] = O.F_POS
        self.pressure_halo = ["pressure", "density"]
        self.count = cap

    def run_reduce(self, phase, lo, hi):
        self.sys.run_phase(phase)
        return self.sys.error_total_fixed(lo, hi)

    def has(self, name):
        return name in self.map

    def set_count(self, n):
        self.sys.set_count(n)
        self.count = n

    def run(self, phase):
        self.sys.run_phase(phase)

    def read(self, name, lo, hi):
        self.sys.set_count(self.cap)               # whole-capacity view for the copy
        out = torch.from_numpy(self.sys.get(self.map[name])[lo:hi].copy())
        self.sys.set_count(self.count)
        return out

    def write(self, name, lo, t):
        self.sys.set_count(self.cap)
        full = self.sys.get(self.map[name])
        full[lo:lo + t.shape[0]] = t.numpy()
        self.sys.set(self.map[name], full)
        self.sys.set_count(self.count)

    def cell_starts(self, idx):
        cs = self.sys.get(self.O.F_CELLSTART_F)
        return [int(cs[i]) for i in idx]

    def columns(self, lo, hi, cell_length):
        return torch.from_numpy(cell_column(self.read("pos", lo, hi).numpy()[:, 0], cell_length))

    def to_device(self, arr):
        return torch.as_tensor(arr)
