"""ctypes binding of the CPU checker (oracle/librebx_oracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_uint8, c_void_p

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_SO = os.path.join(ORACLE_DIR, "librebx_oracle.so")


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR])
    return ctypes.CDLL(_SO)


lib = _load()


class OrcSim(ctypes.Structure):
    _fields_ = [("N", c_int), ("G", c_double)] + [(k, c_void_p) for k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")]


def _d(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class Sim:
    """Owns contiguous float64 copies of the columns; accelerations are updated in place."""

    def __init__(self, cols, G=1.0):
        self.cols = {k: np.ascontiguousarray(np.array(cols[k], dtype=np.float64)) for k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")}
        self.N = len(self.cols["x"])
        self.G = float(G)
        self.s = OrcSim(self.N, self.G, *[_d(self.cols[k]) for k in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")])

    @property
    def acc(self):
        return self.cols["ax"], self.cols["ay"], self.cols["az"]


def _col(a, n):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    assert len(a) == n
    return a


def _flag(a, n):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a).astype(np.uint8))
    assert len(a) == n
    return a


for _f in ("orc_radiation_forces", "orc_gravitational_harmonics", "orc_gr", "orc_gr_literal", "orc_gr_potential", "orc_gr_full",
           "orc_central_force", "orc_tides_precession", "orc_modify_orbits_forces"):
    getattr(lib, _f).restype = c_int
for _f in ("orc_gravitational_harmonics_potential", "orc_central_force_potential", "orc_tides_precession_potential",
           "orc_gr_potential_potential", "orc_gr_hamiltonian", "orc_newtonian_energy", "orc_rad_calc_beta", "orc_rad_calc_particle_radius"):
    getattr(lib, _f).restype = c_double
lib.orc_rad_calc_beta.argtypes = [c_double] * 7
lib.orc_rad_calc_particle_radius.argtypes = [c_double] * 7


def radiation_forces(sim, c, beta=None, radiation_source=None):
    b, f = _col(beta, sim.N), _flag(radiation_source, sim.N)
    return lib.orc_radiation_forces(ctypes.byref(sim.s), c_double(c), _d(b), _d(f))


def gravitational_harmonics(sim, J2=None, J4=None, R_eq=None):
    a, b, c = _col(J2, sim.N), _col(J4, sim.N), _col(R_eq, sim.N)
    return lib.orc_gravitational_harmonics(ctypes.byref(sim.s), _d(a), _d(b), _d(c))


def gr(sim, c, max_iterations=10, literal=False):
    fn = lib.orc_gr_literal if literal else lib.orc_gr
    return fn(ctypes.byref(sim.s), c_double(c), c_int(max_iterations))


def gr_potential(sim, c):
    return lib.orc_gr_potential(ctypes.byref(sim.s), c_double(c))


def gr_full(sim, c, gravity_ignore_10=0):
    return lib.orc_gr_full(ctypes.byref(sim.s), c_double(c), c_int(gravity_ignore_10))


def central_force(sim, Acentral=None, gammacentral=None):
    a, b = _col(Acentral, sim.N), _col(gammacentral, sim.N)
    return lib.orc_central_force(ctypes.byref(sim.s), _d(a), _d(b))


def tides_precession(sim, R_tides=None, k1=None, tides_primary=None):
    a, b, f = _col(R_tides, sim.N), _col(k1, sim.N), _flag(tides_primary, sim.N)
    return lib.orc_tides_precession(ctypes.byref(sim.s), _d(a), _d(b), _d(f))


def modify_orbits_forces(sim, coordinates=0, tau_a=None, tau_e=None, tau_inc=None, primary=None):
    a, b, c, f = _col(tau_a, sim.N), _col(tau_e, sim.N), _col(tau_inc, sim.N), _flag(primary, sim.N)
    return lib.orc_modify_orbits_forces(ctypes.byref(sim.s), c_int(coordinates), _d(a), _d(b), _d(c), _d(f))


def _hp(fn, sim, *args):
    """Calls one of the binary128 evaluations; returns (hi, lo), each (3, N) float64, value = hi + lo."""
    hi, lo = np.zeros((3, sim.N)), np.zeros((3, sim.N))
    ptrs = (c_void_p * 6)(*[hi[k].ctypes.data for k in range(3)], *[lo[k].ctypes.data for k in range(3)])
    rc = fn(ctypes.byref(sim.s), *args, ptrs)
    assert rc == 0
    return hi, lo


def hp_gr(sim, c, max_iterations=10):
    """What gr adds to the accelerations, evaluated in binary128 (oracle/rebx_oracle_hp.c)."""
    return _hp(lib.orc_hp_gr, sim, c_double(c), c_int(max_iterations))


def hp_modify_orbits_forces(sim, coordinates=0, tau_a=None, tau_e=None, tau_inc=None):
    a, b, c = _col(tau_a, sim.N), _col(tau_e, sim.N), _col(tau_inc, sim.N)
    return _hp(lib.orc_hp_modify_orbits_forces, sim, c_int(coordinates), _d(a), _d(b), _d(c))


def hp_force(sim, force, pp):
    """(hi, lo) of one prefix-sum force given as (name, effect parameters)."""
    name, ep = force
    if name == "gr":
        return hp_gr(sim, ep["c"], ep.get("max_iterations", 10))
    if name == "modify_orbits_forces" and ep.get("coordinates", 0) in (0, 1):
        return hp_modify_orbits_forces(sim, ep.get("coordinates", 0), pp.get("tau_a"), pp.get("tau_e"), pp.get("tau_inc"))
    raise ValueError("no binary128 evaluation of %s" % (force,))


def newtonian_gravity(sim):
    lib.orc_newtonian_gravity(ctypes.byref(sim.s))


def reset_accelerations(sim):
    lib.orc_reset_accelerations(ctypes.byref(sim.s))


def harmonics_potential(sim, J2=None, J4=None, R_eq=None):
    a, b, c = _col(J2, sim.N), _col(J4, sim.N), _col(R_eq, sim.N)
    return lib.orc_gravitational_harmonics_potential(ctypes.byref(sim.s), _d(a), _d(b), _d(c))


def central_force_potential(sim, Acentral=None, gammacentral=None):
    a, b = _col(Acentral, sim.N), _col(gammacentral, sim.N)
    return lib.orc_central_force_potential(ctypes.byref(sim.s), _d(a), _d(b))


def tides_potential(sim, R_tides=None, k1=None, primary=None):
    a, b, f = _col(R_tides, sim.N), _col(k1, sim.N), _flag(primary, sim.N)
    return lib.orc_tides_precession_potential(ctypes.byref(sim.s), _d(a), _d(b), _d(f))


def gr_potential_potential(sim, c):
    return lib.orc_gr_potential_potential(ctypes.byref(sim.s), c_double(c))


def gr_hamiltonian(sim, c):
    return lib.orc_gr_hamiltonian(ctypes.byref(sim.s), c_double(c))


lib.orc_gr_full_hamiltonian.restype = c_double


def gr_full_hamiltonian(sim, c):
    return lib.orc_gr_full_hamiltonian(ctypes.byref(sim.s), c_double(c))


def newtonian_energy(sim):
    return lib.orc_newtonian_energy(ctypes.byref(sim.s))


def apply_forces(sim, forces, pp, gravity_ignore_terms=0):
    """Runs a stack of reference forces in EXECUTION order with the checker, one after the
    other, as the reference dispatch does (src/core.c:868-880)."""
    g = lambda k: pp.get(k)
    for name, ep in forces:
        if name == "radiation_forces":
            radiation_forces(sim, ep["c"], g("beta"), g("radiation_source"))
        elif name == "gravitational_harmonics":
            gravitational_harmonics(sim, g("J2"), g("J4"), g("R_eq"))
        elif name == "gr":
            gr(sim, ep["c"], ep.get("max_iterations", 10))
        elif name == "gr_potential":
            gr_potential(sim, ep["c"])
        elif name == "gr_full":
            gr_full(sim, ep["c"], 1 if gravity_ignore_terms == 1 else 0)
        elif name == "central_force":
            central_force(sim, g("Acentral"), g("gammacentral"))
        elif name == "tides_precession":
            tides_precession(sim, g("R_tides"), g("k1"), g("tides_primary"))
        elif name == "modify_orbits_forces":
            rc = modify_orbits_forces(sim, ep.get("coordinates", 0), g("tau_a"), g("tau_e"), g("tau_inc"), g("primary"))
            assert rc == 0
        else:
            raise ValueError(name)


INTEGRATORS = {"none": -1, "implicit_midpoint": 0, "rk4": 1, "euler": 2, "rk2": 3}
_FORCE_FN = ctypes.CFUNCTYPE(None, POINTER(OrcSim), c_void_p)
lib.orc_integrate_force.restype = c_int
lib.orc_integrate_force.argtypes = [POINTER(OrcSim), c_int, c_double, _FORCE_FN, c_void_p]
lib.orc_drift_step.argtypes = [POINTER(OrcSim), c_double]
lib.orc_kick_step.argtypes = [POINTER(OrcSim), c_double]


class _StateView:
    """A stage state handed to the force callback by the checker's integrators."""

    def __init__(self, s):
        self.s, self.N, self.G = s, s.N, s.G


def integrate_force(sim, integrator, dt, forces, pp, gravity_ignore_terms=0):
    """src/integrate_force.c:34-78 with `forces` (execution order) as the integrated force.
    Velocities of `sim` are updated in place; returns the implicit-midpoint loop counter."""
    def cb(ptr, _user):
        apply_forces(_StateView(ptr.contents), forces, pp, gravity_ignore_terms)
    return lib.orc_integrate_force(ctypes.byref(sim.s), INTEGRATORS[integrator] if isinstance(integrator, str) else integrator,
                                   float(dt), _FORCE_FN(cb), None)


def drift_step(sim, dt):
    lib.orc_drift_step(ctypes.byref(sim.s), float(dt))


def kick_step(sim, dt):
    lib.orc_kick_step(ctypes.byref(sim.s), float(dt))


_STEP_FN = ctypes.CFUNCTYPE(None, POINTER(OrcSim), c_double, c_void_p)
lib.orc_whfast_steps.restype = c_int
lib.orc_whfast_steps.argtypes = [POINTER(OrcSim), c_double, ctypes.c_long, _FORCE_FN, _STEP_FN, _STEP_FN, c_void_p]


def whfast_steps(sim, dt, nsteps, forces=None, pp=None, pre=None, post=None):
    """Wisdom-Holman steps (oracle/rebx_whfast.c) with `forces` (execution order) applied in
    the kick.  pre/post: callables (state_view, dt) run before/after every step."""
    def cb(ptr, _user):
        # WHFast leaves the (0,1) pair out of the accelerations it hands to the forces
        # (sim->gravity_ignore_terms == 1): gr_full adds it back (src/gr_full.c:150-171)
        apply_forces(_StateView(ptr.contents), forces, pp or {}, gravity_ignore_terms=1)

    def wrap(fn):
        if fn is None:
            return _STEP_FN(0)
        return _STEP_FN(lambda ptr, h, _u: fn(_StateView(ptr.contents), h))
    f = _FORCE_FN(cb) if forces else _FORCE_FN(0)
    rc = lib.orc_whfast_steps(ctypes.byref(sim.s), float(dt), int(nsteps), f, wrap(pre), wrap(post), None)
    assert rc == 0


lib.orc_modify_mass.restype = None
lib.orc_modify_mass.argtypes = [POINTER(OrcSim), c_void_p, c_double]
lib.orc_modify_orbits_direct.restype = c_int
lib.orc_modify_orbits_direct.argtypes = [POINTER(OrcSim), c_int, c_double] + [c_void_p] * 6 + [c_double]


def modify_mass(sim, dt, tau_mass=None):
    a = _col(tau_mass, sim.N)
    lib.orc_modify_mass(ctypes.byref(sim.s), _d(a), float(dt))


def modify_orbits_direct(sim, dt, coordinates=0, p=None, tau_a=None, tau_e=None, tau_inc=None, tau_omega=None, tau_Omega=None, primary=None):
    cols = [_col(c, sim.N) for c in (tau_a, tau_e, tau_inc, tau_omega, tau_Omega)]
    f = _flag(primary, sim.N)
    return lib.orc_modify_orbits_direct(ctypes.byref(sim.s), int(coordinates), float("nan") if p is None else float(p),
                                        *[_d(c) for c in cols], _d(f), float(dt))
