"""reboundx/test/test_conservation.py re-stated end to end on the device path: for every force that derives from a Hamiltonian,
the trajectory is integrated with Newtonian gravity + the force evaluated ON THE DEVICE through sim->additional_forces, and
the Hamiltonian -- the device's own rebx_*_potential / rebx_gr_hamiltonian / rebx_gr_full_hamiltonian through the plugin API
-- must come out conserved.  The CPU checker takes no part: force and potential are checked against each other.

The reference integrates its three-body system to t = 1e4 with IAS15 (REBOUND, absent here) and asks for 1e-12; here an
8th-order Runge-Kutta scheme at rtol 1e-13 over 100 time units, bar 1e-11 (the same relaxation as the CPU restatement of the
test, tests/test_oracle.py::test_force_conserves_its_hamiltonian)."""
import numpy as np
import pytest

from reboundx_amd import synthetic
from reboundx_amd.extras import Extras, Simulation

pytestmark = pytest.mark.gpu

# parameters of reboundx/test/test_conservation.py
SETUPS = {
    "gr": ({"c": 1e4}, {}),
    "gr_full": ({"c": 1e4}, {}),
    "gr_potential": ({"c": 1e4}, {}),
    "tides_precession": ({}, {"R_tides": 1e-3, "k1": 0.4}),
    "central_force": ({}, {"Acentral": 1e-4, "gammacentral": -1.0}),
    "gravitational_harmonics": ({}, {"J2": 1e-3, "J4": 1e-3, "R_eq": 1e-3}),
}


def newtonian(m, x):
    """accelerations (3, n) and potential energy of point masses, G = 1"""
    n = len(m)
    a = np.zeros((3, n))
    e = 0.0
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            d = x[:, j] - x[:, i]
            r = np.sqrt((d * d).sum())
            a[:, i] += m[j] * d / r ** 3
            if j > i:
                e -= m[i] * m[j] / r
    return a, e


def build(name, mode):
    cols = synthetic.twoplanets()
    sim = Simulation()
    for i in range(3):
        sim.add(m=cols["m"][i], x=cols["x"][i], y=cols["y"][i], z=cols["z"][i], vx=cols["vx"][i], vy=cols["vy"][i], vz=cols["vz"][i])
    rebx = Extras(sim)
    rebx.set_mode(mode)
    force = rebx.load_force(name)
    rebx.add_force(force)
    ep, p0 = SETUPS[name]
    for k, v in ep.items():
        force.params[k] = v
    for k, v in p0.items():
        sim.particles[0].params[k] = v
    return sim, rebx, force, cols["m"].copy()


def hamiltonian(name, sim, rebx, force, m):
    ps = sim.particles
    x = np.array([[ps[i].x, ps[i].y, ps[i].z] for i in range(3)]).T
    v = np.array([[ps[i].vx, ps[i].vy, ps[i].vz] for i in range(3)]).T
    energy = 0.5 * float((m * (v * v).sum(axis=0)).sum()) + newtonian(m, x)[1]
    if name == "gr":
        return rebx.gr_hamiltonian(force), energy
    if name == "gr_full":
        return rebx.gr_full_hamiltonian(force), energy
    pot = {"gr_potential": lambda: rebx.gr_potential_potential(force), "tides_precession": rebx.tides_precession_potential,
           "central_force": rebx.central_force_potential, "gravitational_harmonics": rebx.gravitational_harmonics_potential}[name]()
    return energy + pot, energy


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("name", sorted(SETUPS))
def test_force_on_the_device_conserves_the_hamiltonian_the_device_reports(name, mode):
    from scipy.integrate import solve_ivp
    sim, rebx, force, m = build(name, mode)
    ps = sim.particles
    h0, e0 = hamiltonian(name, sim, rebx, force, m)

    def rhs(t, y):
        q = y.reshape(6, 3)
        a, _ = newtonian(m, q[:3])
        for i in range(3):
            p = ps[i]
            p.x, p.y, p.z, p.vx, p.vy, p.vz = q[:, i]
            p.ax, p.ay, p.az = a[:, i]           # gr_full reads the Newtonian accelerations it finds (src/gr_full.c:70-74)
        sim.call_additional_forces()
        return np.concatenate([q[3], q[4], q[5], [ps[i].ax for i in range(3)], [ps[i].ay for i in range(3)], [ps[i].az for i in range(3)]])

    y0 = np.array([[ps[i].x for i in range(3)], [ps[i].y for i in range(3)], [ps[i].z for i in range(3)],
                   [ps[i].vx for i in range(3)], [ps[i].vy for i in range(3)], [ps[i].vz for i in range(3)]]).ravel()
    sol = solve_ivp(rhs, (0.0, 100.0), y0, method="DOP853", rtol=1e-13, atol=1e-18)
    assert sol.success
    q = sol.y[:, -1].reshape(6, 3)
    for i in range(3):
        ps[i].x, ps[i].y, ps[i].z, ps[i].vx, ps[i].vy, ps[i].vz = q[:, i]
    h1, e1 = hamiltonian(name, sim, rebx, force, m)
    assert abs((h1 - h0) / h0) < 1e-11, (h0, h1)
    assert np.abs(q - y0.reshape(6, 3)).max() > 0.1          # the system has moved
    if name in ("gr", "gr_full", "gr_potential"):
        assert abs((e1 - e0) / e0) > 1e-10                   # the Newtonian energy alone is not conserved: the force acted
    sim.free_particle_params()
