"""Loads librebxamd.so and declares the C ABI (include/reboundx_amd.h, group B)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librebxamd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "reboundx_amd: %s is missing. Build it first: python -c 'import __graft_entry__ as g; g.build()' "
        "(or make -C reboundx_amd/csrc). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

c_double_p = POINTER(c_double)


class Op(ctypes.Structure):
    """struct rebxamd_op"""
    _fields_ = [("kind", c_int32), ("src", c_int32), ("iarg", c_int32), ("reserved", c_int32),
                ("a", c_double), ("b", c_double)]


class View(ctypes.Structure):
    """struct rebxamd_view (device pointers as integers)"""
    _fields_ = [("n", c_int64), ("n_head", c_int64), ("tail_massless", c_int32), ("reserved", c_int32),
                ("G", c_double),
                ("x", c_void_p), ("y", c_void_p), ("z", c_void_p),
                ("vx", c_void_p), ("vy", c_void_p), ("vz", c_void_p),
                ("m", c_void_p),
                ("ax", c_void_p), ("ay", c_void_p), ("az", c_void_p),
                ("beta", c_void_p), ("R_tides", c_void_p), ("k1", c_void_p),
                ("tau_a", c_void_p), ("tau_e", c_void_p), ("tau_inc", c_void_p)]


OP_RADIATION, OP_J2, OP_J4, OP_GR, OP_GR_POTENTIAL, OP_CENTRAL, OP_TIDES, OP_MOF, OP_GR_FULL = range(1, 10)
MODE_FAST, MODE_EXACT = 0, 1
COORD_JACOBI, COORD_BARYCENTRIC, COORD_PARTICLE = 0, 1, 2
OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_PARAM = range(5)
MAX_OPS, MAX_HEAD = 32, 64
WARN_GR_ITER, WARN_GRFULL_SWEEPS, WARN_IM_ITER = 1, 2, 4
INTEGRATOR_NONE, INTEGRATOR_IMPLICIT_MIDPOINT, INTEGRATOR_RK4, INTEGRATOR_EULER, INTEGRATOR_RK2 = -1, 0, 1, 2, 3

_sig = {
    "rebxamd_create": (c_int, [c_int, POINTER(c_void_p)]),
    "rebxamd_destroy": (None, [c_void_p]),
    "rebxamd_last_error": (c_char_p, [c_void_p]),
    "rebxamd_classify": (c_int, [c_void_p, c_void_p, c_int64, POINTER(c_int64), c_void_p]),
    "rebxamd_apply": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_int, c_void_p]),
    "rebxamd_apply_static": (c_int, [POINTER(View), POINTER(Op), c_int, c_int, c_void_p, c_void_p, c_char_p, c_int]),
    "rebxamd_apply_targets": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_int, c_int64, c_void_p, c_void_p]),
    "rebxamd_add_pushes": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_void_p, c_void_p]),
    "rebxamd_shard_sums": (c_int, [c_void_p, POINTER(View), c_int64, c_int64, c_void_p, c_void_p]),
    "rebxamd_shard_mof_forces": (c_int, [c_void_p, POINTER(View), c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "rebxamd_shard_mof_apply": (c_int, [c_void_p, POINTER(View), c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "rebxamd_shard_massive": (c_int, [c_void_p, POINTER(View), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "rebxamd_shard_gr_newton": (c_int, [c_void_p, POINTER(View), c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "rebxamd_shard_gr_forces": (c_int, [c_void_p, POINTER(View), c_double, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "rebxamd_shard_gr_apply": (c_int, [c_void_p, POINTER(View), c_int64, c_int64, c_void_p, c_void_p]),
    "rebxamd_reset_accelerations": (c_int, [c_void_p, POINTER(View), c_void_p]),
    "rebxamd_wh_drift": (c_int, [c_void_p, POINTER(View), c_double, c_int, c_void_p]),
    "rebxamd_wh_steps": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_double, c_int64, c_int, c_void_p]),
    "rebxamd_wh_run_begin": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_double, c_int, c_void_p]),
    "rebxamd_wh_run_steps": (c_int, [c_void_p, c_int64, c_void_p]),
    "rebxamd_wh_run_end": (c_int, [c_void_p, c_void_p]),
    "rebxamd_modify_mass": (c_int, [c_void_p, POINTER(View), c_void_p, c_double, c_void_p]),
    "rebxamd_modify_orbits_direct": (c_int, [c_void_p, POINTER(View), c_int, c_int, POINTER(c_double), c_void_p, c_void_p, c_double, c_void_p]),
    "rebxamd_potentials": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, POINTER(c_double), c_void_p]),
    "rebxamd_gr_hamiltonian": (c_int, [c_void_p, POINTER(View), c_double, POINTER(c_double), c_void_p]),
    "rebxamd_gr_full_hamiltonian": (c_int, [c_void_p, POINTER(View), c_double, POINTER(c_double), c_void_p]),
    "rebxamd_integrate_force": (c_int, [c_void_p, POINTER(View), POINTER(Op), c_int, c_int, c_double, c_int, POINTER(c_int), c_void_p]),
    "rebxamd_kick": (c_int, [c_void_p, POINTER(View), c_double, c_int, c_void_p]),
    "rebxamd_drift": (c_int, [c_void_p, POINTER(View), c_double, c_int, c_void_p]),
    "rebxamd_fetch_warnings": (c_int, [c_void_p, POINTER(c_uint32), c_void_p]),
    "rebxamd_timing_enable": (c_int, [c_void_p, c_int]),
    "rebxamd_direct_chain": (c_int, [c_void_p, c_int, POINTER(c_int)]),
    "rebxamd_fixed_stacks": (c_int, [c_void_p, c_int]),
    "rebxamd_carried_steps": (c_int, [c_void_p, c_int]),
    "rebxamd_sequential_exact": (c_int, [c_void_p, c_int]),
    "rebxamd_grfull_one_launch": (c_int, [c_void_p, c_int]),
    "rebxamd_wh_batch": (c_int, [c_void_p, c_int]),
    "rebxamd_timing_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    "rebxamd_grfull_newton": (c_int, [c_void_p, POINTER(View), c_int] + [c_void_p] * 6 + [c_void_p]),
    "rebxamd_grfull_finish": (c_int, [c_void_p, POINTER(View), c_void_p, c_void_p, c_void_p, c_void_p]),
    "rebxamd_grfull_pot": (c_int, [c_void_p, POINTER(View), c_double, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "rebxamd_grfull_const": (c_int, [c_void_p, POINTER(View), c_double, c_int64, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rebxamd_grfull_sweep": (c_int, [c_void_p, POINTER(View), c_double, c_int64, c_int64] + [c_void_p] * 13 + [c_int, c_void_p]),
    "rebxamd_grfull_pot_sweep0": (c_int, [c_void_p, POINTER(View), c_double, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rebxamd_grfull_sweep0_finish": (c_int, [c_void_p, POINTER(View), c_double, c_int64, c_int64] + [c_void_p] * 13 + [c_int, c_void_p]),
    "rebxamd_grfull_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "rebxamd_grfull_unpack": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rebxamd_grfull_stop_word": (c_int, [c_void_p, c_void_p]),
    "rebxamd_grfull": (c_int, [c_void_p, POINTER(View), c_double, c_int, c_int, POINTER(c_int), c_void_p]),
}
for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

DEVICE_API_SYMBOLS = tuple(_sig)
