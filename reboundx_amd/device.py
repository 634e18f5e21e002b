"""Device-resident particles and the thin wrapper over the device C ABI.

PyTorch is used only for HBM allocations and streams; every computation goes through
``librebxamd.so``.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import Op, View, lib
from .program import build_ops

DYN = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
PARAM_COLS = ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc")


class RebxAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librebxamd error %d: %s" % (code, msg))
        self.code = code


class Context:
    """One per process and GPU (struct rebxamd_ctx)."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("reboundx_amd.device needs a GPU; there is no CPU fallback")
        self.device = int(device)
        h = ctypes.c_void_p()
        rc = lib.rebxamd_create(self.device, ctypes.byref(h))
        if rc != _lib.OK:
            raise RebxAmdError(rc, "rebxamd_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.rebxamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.OK:
            raise RebxAmdError(rc, lib.rebxamd_last_error(self._h).decode())

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def classify(self, m):
        out = ctypes.c_int64(0)
        self._check(lib.rebxamd_classify(self._h, ctypes.c_void_p(m.data_ptr()), m.numel(), ctypes.byref(out), self._stream()))
        return out.value

    def apply(self, view, ops, mode=_lib.MODE_FAST):
        arr = (Op * max(len(ops), 1))(*ops)
        self._check(lib.rebxamd_apply(self._h, ctypes.byref(view), arr, len(ops), mode, self._stream()))

    def apply_targets(self, view, ops, target_lo, sums, mode=_lib.MODE_FAST):
        arr = (Op * len(ops))(*ops)
        self._check(lib.rebxamd_apply_targets(self._h, ctypes.byref(view), arr, len(ops), mode, target_lo, sums.data_ptr(), self._stream()))

    def add_pushes(self, view, ops, sums):
        arr = (Op * len(ops))(*ops)
        self._check(lib.rebxamd_add_pushes(self._h, ctypes.byref(view), arr, len(ops), sums.data_ptr(), self._stream()))

    def wh_drift(self, view, dt, mode=_lib.MODE_FAST):
        self._check(lib.rebxamd_wh_drift(self._h, ctypes.byref(view), float(dt), mode, self._stream()))

    def wh_steps(self, view, ops, dt, nsteps, mode=_lib.MODE_FAST):
        """nsteps Wisdom-Holman steps (one dominant mass + massless bodies) with `ops` in the kick."""
        arr = (Op * max(len(ops), 1))(*ops)
        self._check(lib.rebxamd_wh_steps(self._h, ctypes.byref(view), arr, len(ops), float(dt), int(nsteps), mode, self._stream()))

    def wh_run_begin(self, view, ops, dt, mode=_lib.MODE_FAST):
        """Opens a Wisdom-Holman run that stays unsynchronised between calls (one pass and one Kepler drift per step)."""
        arr = (Op * max(len(ops), 1))(*ops)
        self._check(lib.rebxamd_wh_run_begin(self._h, ctypes.byref(view), arr, len(ops), float(dt), mode, self._stream()))

    def wh_run_steps(self, nsteps):
        self._check(lib.rebxamd_wh_run_steps(self._h, int(nsteps), self._stream()))

    def wh_run_end(self):
        """Closing half drift: the columns hold the synchronised state."""
        self._check(lib.rebxamd_wh_run_end(self._h, self._stream()))

    def modify_mass(self, view, tau_mass, dt):
        """tau_mass: float64 device tensor, NaN = not set.  Masses, positions, velocities change in place."""
        self._check(lib.rebxamd_modify_mass(self._h, ctypes.byref(view), tau_mass.data_ptr(), float(dt), self._stream()))

    def modify_orbits_direct(self, view, dt, coordinates=0, primary=-1, p=None, tau_omega=None, tau_Omega=None):
        pp = ctypes.byref(ctypes.c_double(p)) if p is not None else None
        self._check(lib.rebxamd_modify_orbits_direct(self._h, ctypes.byref(view), int(coordinates), int(primary), pp,
                                                     tau_omega.data_ptr() if tau_omega is not None else None,
                                                     tau_Omega.data_ptr() if tau_Omega is not None else None, float(dt), self._stream()))

    def direct_chain(self, max_rounds=-1):
        """modify_orbits_direct on layouts with massive bodies anywhere: sets the rounds of the fixed-point iteration tried
        before the sequential walk (-1: leave as is; 0: always walk) and returns the rounds the last such call took (-1: walked)."""
        last = ctypes.c_int(0)
        self._check(lib.rebxamd_direct_chain(self._h, int(max_rounds), ctypes.byref(last)))
        return last.value

    def potentials(self, view, ops):
        """Potential energy of each op (J2, J4, central_force, tides_precession, gr_potential)."""
        arr = (Op * max(len(ops), 1))(*ops)
        out = (ctypes.c_double * max(len(ops), 1))()
        self._check(lib.rebxamd_potentials(self._h, ctypes.byref(view), arr, len(ops), out, self._stream()))
        return [out[k] for k in range(len(ops))]

    def gr_hamiltonian(self, view, c):
        out = ctypes.c_double(0.)
        self._check(lib.rebxamd_gr_hamiltonian(self._h, ctypes.byref(view), float(c), ctypes.byref(out), self._stream()))
        return out.value

    def gr_full_hamiltonian(self, view, c):
        out = ctypes.c_double(0.)
        self._check(lib.rebxamd_gr_full_hamiltonian(self._h, ctypes.byref(view), float(c), ctypes.byref(out), self._stream()))
        return out.value

    def integrate_force(self, view, ops, integrator, dt, mode=_lib.MODE_FAST):
        """rebx_integrate_force on device columns: velocities of `view` are updated in place.
        `integrator`: one of "euler", "rk2", "rk4", "implicit_midpoint", "none" (or the enum
        value).  Returns the implicit-midpoint loop counter."""
        code = {"none": -1, "implicit_midpoint": 0, "rk4": 1, "euler": 2, "rk2": 3}.get(integrator, integrator)
        arr = (Op * max(len(ops), 1))(*ops)
        it = ctypes.c_int(0)
        self._check(lib.rebxamd_integrate_force(self._h, ctypes.byref(view), arr, len(ops), int(code), float(dt), mode,
                                                ctypes.byref(it), self._stream()))
        return it.value

    def kick(self, view, dt, mode=_lib.MODE_FAST):
        self._check(lib.rebxamd_kick(self._h, ctypes.byref(view), float(dt), mode, self._stream()))

    def drift(self, view, dt, mode=_lib.MODE_FAST):
        self._check(lib.rebxamd_drift(self._h, ctypes.byref(view), float(dt), mode, self._stream()))

    def shard_sums(self, view, lo, hi, out7):
        self._check(lib.rebxamd_shard_sums(self._h, ctypes.byref(view), lo, hi, out7.data_ptr(), self._stream()))

    def shard_mof_forces(self, view, coordinates, lo, hi, before7, push3):
        self._check(lib.rebxamd_shard_mof_forces(self._h, ctypes.byref(view), coordinates, lo, hi,
                                                 before7.data_ptr() if before7 is not None else None, push3.data_ptr(), self._stream()))

    def shard_mof_apply(self, view, coordinates, lo, hi, after3):
        self._check(lib.rebxamd_shard_mof_apply(self._h, ctypes.byref(view), coordinates, lo, hi,
                                                after3.data_ptr() if after3 is not None else None, self._stream()))

    def shard_massive(self, view, lo, hi, id0, records, count):
        self._check(lib.rebxamd_shard_massive(self._h, ctypes.byref(view), lo, hi, id0, records.data_ptr(), count.data_ptr(), self._stream()))

    def shard_gr_newton(self, view, lo, hi, id0, records, nrecords, sums10, mode=0):
        self._check(lib.rebxamd_shard_gr_newton(self._h, ctypes.byref(view), lo, hi, id0, records.data_ptr() if nrecords else None, nrecords,
                                                sums10.data_ptr(), mode, self._stream()))

    def shard_gr_forces(self, view, c, max_iterations, lo, hi, before10, push3, mode=0):
        self._check(lib.rebxamd_shard_gr_forces(self._h, ctypes.byref(view), c, max_iterations, lo, hi,
                                                before10.data_ptr() if before10 is not None else None, push3.data_ptr(), mode, self._stream()))

    def shard_gr_apply(self, view, lo, hi, after3):
        self._check(lib.rebxamd_shard_gr_apply(self._h, ctypes.byref(view), lo, hi, after3.data_ptr() if after3 is not None else None, self._stream()))

    def reset_accelerations(self, view):
        self._check(lib.rebxamd_reset_accelerations(self._h, ctypes.byref(view), self._stream()))

    def fetch_warnings(self):
        bits = ctypes.c_uint32(0)
        self._check(lib.rebxamd_fetch_warnings(self._h, ctypes.byref(bits), self._stream()))
        return bits.value

    def timing_enable(self, on=True):
        self._check(lib.rebxamd_timing_enable(self._h, 1 if on else 0))

    def fixed_stacks(self, on=True):
        """Whether stacks with a kernel instance of their own use it (default) or walk the op list."""
        self._check(lib.rebxamd_fixed_stacks(self._h, 1 if on else 0))

    def grfull_one_launch(self, on=True):
        """gr_full on at most 256 bodies: the whole evaluation in one launch (default), or the queued passes (on=False)."""
        self._check(lib.rebxamd_grfull_one_launch(self._h, 1 if on else 0))

    def sequential_exact(self, on=True):
        """Reference-order arithmetic on layouts with massive bodies anywhere: walk gr and Jacobi/barycentric
        modify_orbits_forces in the reference's order up to REBXAMD_SEQ_MAX_BODIES particles (default), or always
        use the parallel prefix sums (on=False)."""
        self._check(lib.rebxamd_sequential_exact(self._h, 1 if on else 0))

    def carried_steps(self, on=True):
        """Whether force passes carry the elementwise passes around them (kick and drift in wh_steps, velocity updates in
        integrate_force, the last pass of a prefix-sum force in apply; default) or every pass runs on its own."""
        self._check(lib.rebxamd_carried_steps(self._h, 1 if on else 0))

    def wh_batch(self, steps_per_pass=32):
        """Steps that one pass of wh_steps carries for every massless body behind a single massive one (1: one pass per step)."""
        self._check(lib.rebxamd_wh_batch(self._h, int(steps_per_pass)))

    def timing_read(self):
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        self._check(lib.rebxamd_timing_read(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def grfull(self, view, c, gravity_ignore_10=0, mode=_lib.MODE_FAST):
        sweeps = ctypes.c_int(0)
        self._check(lib.rebxamd_grfull(self._h, ctypes.byref(view), c, gravity_ignore_10, mode, ctypes.byref(sweeps), self._stream()))
        return sweeps.value

    def grfull_stop_word(self, word):
        """word: int32 device tensor (kept alive by the caller) or None."""
        self._check(lib.rebxamd_grfull_stop_word(self._h, ctypes.c_void_p(word.data_ptr()) if word is not None else None))

    def grfull_newton(self, view, gravity_ignore_10, an, a0):
        self._check(lib.rebxamd_grfull_newton(self._h, ctypes.byref(view), gravity_ignore_10, an[0].data_ptr(), an[1].data_ptr(), an[2].data_ptr(),
                                              a0[0].data_ptr(), a0[1].data_ptr(), a0[2].data_ptr(), self._stream()))

    def grfull_finish(self, view, a):
        self._check(lib.rebxamd_grfull_finish(self._h, ctypes.byref(view), a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), self._stream()))

    def grfull_pot(self, view, c, row0, row1, pot, mode=_lib.MODE_FAST):
        self._check(lib.rebxamd_grfull_pot(self._h, ctypes.byref(view), c, row0, row1, pot.data_ptr(), mode, self._stream()))

    def grfull_const(self, view, c, row0, row1, pot, ac, mode=_lib.MODE_FAST):
        self._check(lib.rebxamd_grfull_const(self._h, ctypes.byref(view), c, row0, row1, pot.data_ptr(),
                                             ac[0].data_ptr(), ac[1].data_ptr(), ac[2].data_ptr(), mode, self._stream()))

    def grfull_pot_sweep0(self, view, c, row0, row1, an, pot, mode=_lib.MODE_FAST):
        """Potential sums of the rows and, in the same pass, the pair sums of the first sweep (kept in the context)."""
        self._check(lib.rebxamd_grfull_pot_sweep0(self._h, ctypes.byref(view), c, row0, row1, an[0].data_ptr(), an[1].data_ptr(), an[2].data_ptr(),
                                                  pot.data_ptr(), mode, self._stream()))

    def grfull_sweep0_finish(self, view, c, row0, row1, an, ao, ac, out, flag, mode=_lib.MODE_FAST):
        ptrs = [t.data_ptr() for t in (an[0], an[1], an[2], ao[0], ao[1], ao[2], ac[0], ac[1], ac[2], out[0], out[1], out[2])]
        self._check(lib.rebxamd_grfull_sweep0_finish(self._h, ctypes.byref(view), c, row0, row1, *ptrs, flag.data_ptr(), mode, self._stream()))

    def grfull_pack(self, out, row0, per, n, flag, send):
        self._check(lib.rebxamd_grfull_pack(self._h, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), row0, per, n, flag.data_ptr(), send.data_ptr(), self._stream()))

    def grfull_unpack(self, recv, world, per, out, flag, stop, vote):
        self._check(lib.rebxamd_grfull_unpack(self._h, recv.data_ptr() if recv is not None else None, world, per,
                                              out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), flag.data_ptr() if flag is not None else None,
                                              stop.data_ptr(), vote.data_ptr(), self._stream()))

    def grfull_sweep(self, view, c, row0, row1, an, ao, ac, out, flag, mode=_lib.MODE_FAST):
        ptrs = [t.data_ptr() for t in (an[0], an[1], an[2], ao[0], ao[1], ao[2], ac[0], ac[1], ac[2], out[0], out[1], out[2])]
        self._check(lib.rebxamd_grfull_sweep(self._h, ctypes.byref(view), c, row0, row1, *ptrs, flag.data_ptr(), mode, self._stream()))


class DeviceParticles:
    """fp64 columns in HBM.  ``cols`` maps names to 1-D float64 tensors on one GPU;
    parameter columns use NaN for 'not set' and may be absent."""

    def __init__(self, cols, G=1.0, n_head=None, tail_massless=False):
        self.cols = cols
        self.G = float(G)
        self.n = int(cols["x"].numel())
        self.n_head = self.n if n_head is None else int(n_head)
        self.tail_massless = bool(tail_massless)
        for k, t in cols.items():
            assert t.dtype == torch.float64 and t.is_contiguous() and t.numel() == self.n, k

    @classmethod
    def from_host(cls, host, device="cuda:0", G=1.0, n_head=None, tail_massless=None):
        """host: dict of numpy columns (dynamic + any of PARAM_COLS)."""
        cols = {}
        for k in DYN + PARAM_COLS:
            if k in host and host[k] is not None:
                cols[k] = torch.from_numpy(np.ascontiguousarray(host[k], dtype=np.float64)).to(device)
        m = np.asarray(host["m"])
        if n_head is None:
            nz = np.nonzero(m != 0.0)[0]
            n_head = int(nz[-1]) + 1 if len(nz) else 1
        if tail_massless is None:
            tail_massless = bool(np.all(m[n_head:] == 0.0))
        return cls(cols, G=G, n_head=n_head, tail_massless=tail_massless)

    def head_view(self):
        """View of the head particles alone (same columns, n = n_head)."""
        v = self.view()
        v.n = min(self.n_head, self.n)
        v.n_head = v.n
        v.tail_massless = 1
        return v

    def view(self):
        v = View()
        v.n = self.n
        v.n_head = self.n_head
        v.tail_massless = 1 if self.tail_massless else 0
        v.G = self.G
        for k in DYN + PARAM_COLS:
            t = self.cols.get(k)
            setattr(v, k, t.data_ptr() if t is not None else None)
        return v

    def acc_host(self):
        return tuple(self.cols[k].cpu().numpy() for k in ("ax", "ay", "az"))


def apply_forces(ctx, dp, forces, pp_host, mode=_lib.MODE_FAST, gravity_ignore_terms=0):
    """Runs a stack of reference forces (execution order) on device-resident particles."""
    ops = build_ops(forces, pp_host, gravity_ignore_terms)
    ctx.apply(dp.view(), ops, mode)
    return ops
