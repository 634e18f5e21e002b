"""Host-side behaviour of the plugin API (no GPU needed): object model, parameter store,
force factory and error channel, after the reference's reboundx/test/test_effects.py,
test_params.py and test_segfaults.py."""
import ctypes
import os
import re

import pytest

from reboundx_amd import LIB_PATH, _lib
from reboundx_amd.extras import PLUGIN_API_SYMBOLS, Extras, Force, Simulation, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_sim():
    sim = Simulation()
    sim.add(m=1.)
    sim.add(m=1e-3, x=1., vy=1.)
    return sim


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "reboundx_amd.h")).read()
    header += open(os.path.join(ROOT, "include", "rebound_min.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rebx_[a-z0-9_]+|rebxamd_[a-z0-9_]+|reb_error|reb_warning)\s*\(", header))
    declared |= set(re.findall(r"extern const char\* (rebx_[a-z_]+);", header))
    declared -= {"rebxamd_ctx", "rebxamd_op", "rebxamd_view"}
    dll = ctypes.CDLL(LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(dll, s)]
    assert not missing, missing
    assert len(declared) > 50
    for s in PLUGIN_API_SYMBOLS + _lib.DEVICE_API_SYMBOLS:
        assert hasattr(dll, s), s


def test_version_strings():
    assert ctypes.c_char_p.in_dll(lib, "rebx_version_str").value == b"3.0.4"
    assert ctypes.c_char_p.in_dll(lib, "rebx_build_str").value


def test_attach_sets_hooks_and_detach_restores_only_its_own():
    sim = make_sim()
    rebx = Extras(sim)
    assert sim.extras == ctypes.addressof(rebx)
    assert sim.free_particle_ap and sim.extras_cleanup
    assert not sim.additional_forces               # installed by the first add_force (core.c:406)
    gr = rebx.load_force("gr")
    rebx.add_force(gr)
    assert sim.additional_forces
    assert sim.force_is_velocity_dependent == 1     # gr is a velocity-dependent force (core.c:392-394)
    rebx.detach(sim)
    assert not sim.additional_forces and not sim.free_particle_ap


def test_load_force_unknown_name_is_an_error_and_leaves_nothing_behind():
    sim = make_sim()
    rebx = Extras(sim)
    with pytest.raises(RuntimeError, match="not found in REBOUNDx library"):
        rebx.load_force("no_such_effect")
    assert lib.rebx_len(rebx._allocated_forces) == 0


def test_every_builtin_force_loads_with_the_reference_force_type():
    sim = make_sim()
    rebx = Extras(sim)
    expected = {"gr": 2, "gr_full": 2, "gr_potential": 1, "radiation_forces": 2, "modify_orbits_forces": 2,
                "tides_precession": 1, "central_force": 1, "gravitational_harmonics": 1}   # core.c:232-263
    for name, ftype in expected.items():
        f = rebx.load_force(name)
        assert f.force_type == ftype and f.name == name.encode() and f.update_accelerations
    assert lib.rebx_len(rebx._allocated_forces) == len(expected)
    assert rebx.get_force("gr_full").name == b"gr_full"
    with pytest.raises(AttributeError):
        rebx.get_force("nope")


def test_add_force_requires_callback_and_type():
    sim = make_sim()
    rebx = Extras(sim)
    f = rebx.create_force("mine")
    with pytest.raises(RuntimeError, match="update_accelerations"):
        rebx.add_force(f)
    f.update_accelerations = lambda s, fo, p, n: None
    with pytest.raises(RuntimeError, match="force_type"):
        rebx.add_force(f)
    f.force_type = "pos"
    rebx.add_force(f)
    assert lib.rebx_len(rebx._additional_forces) == 1
    assert sim.force_is_velocity_dependent == 0


def test_forces_are_listed_in_reverse_order_of_addition():
    sim = make_sim()
    rebx = Extras(sim)
    names = ["gr", "radiation_forces", "tides_precession"]
    for n in names:
        rebx.add_force(rebx.load_force(n))
    node, seen = rebx._additional_forces, []
    while node:
        seen.append(ctypes.cast(node.contents.object, ctypes.POINTER(Force)).contents.name.decode())
        node = node.contents.next
    assert seen == names[::-1]                      # head insertion, linkedlist.c:33-36


def test_remove_force():
    sim = make_sim()
    rebx = Extras(sim)
    a, b = rebx.load_force("gr"), rebx.load_force("gr_potential")
    rebx.add_force(a)
    rebx.add_force(b)
    rebx.remove_force(a)
    assert lib.rebx_len(rebx._additional_forces) == 1 and lib.rebx_len(rebx._allocated_forces) == 1
    c = rebx.load_force("central_force")            # loaded but never added: removal reports failure (core.c:659-661)
    with pytest.raises(AttributeError):
        rebx.remove_force(c)


def test_params_set_get_update_len():
    sim = make_sim()
    rebx = Extras(sim)
    ps = sim.particles
    gr = rebx.load_force("gr")
    gr.params["c"] = 1e4
    gr.params["max_iterations"] = 7
    assert gr.params["c"] == 1e4 and gr.params["max_iterations"] == 7 and len(gr.params) == 2
    gr.params["c"] = 3e4                            # update in place: no new node
    assert gr.params["c"] == 3e4 and len(gr.params) == 2
    ps[1].params["beta"] = 0.25
    ps[1].params["tau_a"] = -1e4
    ps[0].params["radiation_source"] = 1
    assert ps[1].params["beta"] == 0.25 and len(ps[1].params) == 2 and len(ps[0].params) == 1
    with pytest.raises(AttributeError, match="not found on object"):
        ps[0].params["beta"]
    sim.free_particle_params()
    assert len(ps[1].params) == 0


def test_unregistered_and_duplicate_parameter_names():
    sim = make_sim()
    rebx = Extras(sim)
    with pytest.raises(AttributeError, match="register"):
        sim.particles[0].params["my_param"] = 1.0
    rebx.register_param("my_param", "REBX_TYPE_DOUBLE")
    sim.particles[0].params["my_param"] = 2.5
    assert sim.particles[0].params["my_param"] == 2.5
    with pytest.raises(RuntimeError, match="already in registered list"):
        rebx.register_param("my_param", "REBX_TYPE_INT")
    with pytest.raises(RuntimeError, match="already in registered list"):
        rebx.register_param("beta", "REBX_TYPE_DOUBLE")
    # the C setter reports an unregistered name and changes nothing (core.c:518-524)
    ap = ctypes.c_void_p()
    lib.rebx_set_param_double(ctypes.byref(rebx), ctypes.byref(ap), b"other", 1.0)
    with pytest.raises(RuntimeError, match="Need to register parameter name 'other'"):
        sim.process_messages()
    assert not ap.value


def test_parameter_column_is_stored_whole_and_rejects_names_it_cannot_hold():
    """rebxamd_set_param_column: the bulk form of rebx_set_param_double (core.c:552-565) for the per-particle
    doubles the built-in effects read.  Host behaviour only: what is kept, what is refused."""
    import numpy as np
    sim = make_sim()
    rebx = Extras(sim)
    assert rebx.get_param_column("tau_a") is None
    col = np.array([np.nan, -1e4])
    assert rebx.set_param_column("tau_a", col)
    col[1] = 5.                                     # the values were copied
    got = rebx.get_param_column("tau_a")
    assert np.isnan(got[0]) and got[1] == -1e4 and len(got) == 2
    assert len(sim.particles[1].params) == 0        # no list node was made
    assert rebx.set_param_column("tau_a", np.array([1., 2., 3.]))       # replaced whole
    assert list(rebx.get_param_column("tau_a")) == [1., 2., 3.]
    assert rebx.set_param_column("tau_a", None)
    assert rebx.get_param_column("tau_a") is None
    with pytest.raises(RuntimeError, match="can be given as a column"):
        rebx.set_param_column("J2", np.array([1e-3, np.nan]))           # a source parameter: stays in the lists
    with pytest.raises(RuntimeError, match="can be given as a column"):
        rebx.set_param_column("primary", np.array([1., np.nan]))
    # an extras object whose registry lacks the name refuses it, as the per-particle setter does (core.c:518-524)
    bare = Simulation()
    bare.add(m=1.)
    raw = Extras.__new__(Extras)
    lib.rebx_initialize(ctypes.byref(bare), ctypes.byref(raw))
    ok = lib.rebxamd_set_param_column(ctypes.byref(raw), b"beta", col.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 2)
    assert ok == 0
    with pytest.raises(RuntimeError, match="Need to register parameter name 'beta'"):
        bare.process_messages()
    lib.rebx_free_pointers(ctypes.byref(raw))


def test_params_move_with_particles_and_pointer_params_are_borrowed():
    sim = make_sim()
    rebx = Extras(sim)
    sim.particles[1].params["beta"] = 0.5
    sim.add(m=0., x=2.)                             # reallocates the particle array
    assert sim.particles[1].params["beta"] == 0.5
    op = rebx.load_force("gr")
    host = rebx.load_force("gr_potential")
    host.params["force"] = op
    assert host.params["force"].name == b"gr"


def test_attach_and_free_through_the_c_entry_points_repeatedly():
    for _ in range(50):                             # test_segfaults.py: create/free cycles
        sim = make_sim()
        rebx = lib.rebx_attach(ctypes.byref(sim))
        f = lib.rebx_load_force(rebx, b"gr")
        lib.rebx_add_force(rebx, f)
        ap = ctypes.c_void_p()
        lib.rebx_set_param_double(rebx, ctypes.byref(ap), b"c", 1e4)
        sim.particles[1].params["beta"] = 1.0
        sim.free_particle_params()
        lib.rebx_free(rebx)
        assert not sim.additional_forces


def test_rad_calc_beta_matches_reference_notebook():
    sim = make_sim()
    rebx = Extras(sim)
    assert rebx.rad_calc_beta(6.674e-11, 3.e8, 1.99e30, 3.85e26, 1.e-5, 1000., 1.) == pytest.approx(0.05767021830984005, rel=1e-15)


# ------------------------------------------------------------------ operators (host side)

def test_operator_factory_and_step_lists():
    """rebx_load_operator / rebx_add_operator follow src/core.c:316-368, :462-506: with WHFast a
    half step before and a half step after the timestep; with IAS15 one whole step after."""
    from reboundx_amd.extras import Operator
    sim = Simulation(integrator="whfast")
    sim.add(m=1.)
    sim.add(m=1e-3, x=1., vy=1.)
    sim.dt = 0.5
    rebx = Extras(sim)
    with pytest.raises(RuntimeError, match="Operator 'kepler' not found"):
        rebx.load_operator("kepler")
    assert lib.rebx_len(rebx._allocated_operators) == 0
    drift = rebx.load_operator("drift")
    assert isinstance(drift, Operator) and drift.name == b"drift" and drift.operator_type == 1
    rebx.add_operator(drift)
    assert lib.rebx_len(rebx._pre_timestep_modifications) == 1 and lib.rebx_len(rebx._post_timestep_modifications) == 1
    assert sim.pre_timestep_modifications and sim.post_timestep_modifications
    sim.pre_timestep_modifications(ctypes.byref(sim))            # what REBOUND does before a step
    assert sim.particles[1].y == pytest.approx(0.25)             # dt/2 * vy
    sim.post_timestep_modifications(ctypes.byref(sim))
    assert sim.particles[1].y == pytest.approx(0.5)
    assert rebx.get_operator("drift").name == b"drift"
    rebx.remove_operator(drift)
    assert lib.rebx_len(rebx._pre_timestep_modifications) == 0 and lib.rebx_len(rebx._post_timestep_modifications) == 0
    with pytest.raises(AttributeError):
        rebx.get_operator("drift")

    sim2 = Simulation(integrator="ias15")
    sim2.add(m=1.)
    rebx2 = Extras(sim2)
    kick = rebx2.load_operator("kick")
    rebx2.add_operator(kick)
    assert lib.rebx_len(rebx2._pre_timestep_modifications) == 0 and lib.rebx_len(rebx2._post_timestep_modifications) == 1
    rebx2.add_operator(kick, dtfraction=0.25, timing="pre")
    assert lib.rebx_len(rebx2._pre_timestep_modifications) == 1


def test_track_min_distance_records_the_closest_approach():
    """src/track_min_distance.c:66-96: a recorder (one whole step after the timestep, core.c:474-479) that lowers a
    particle's "min_distance" to its distance from particle 0, or from the particle whose hash "min_distance_from" holds,
    whenever that is smaller, and never raises it."""
    sim = Simulation(integrator="whfast")
    sim.add(m=1.)
    sim.add(m=1e-3, x=3., y=4.)
    sim.add(m=1e-3, x=3., y=4., z=12.)
    sim.particles[2].hash = 777
    rebx = Extras(sim)
    op = rebx.load_operator("track_min_distance")
    assert op.operator_type == 2                                  # REBX_OPERATOR_RECORDER
    rebx.add_operator(op)
    assert lib.rebx_len(rebx._pre_timestep_modifications) == 0 and lib.rebx_len(rebx._post_timestep_modifications) == 1
    ps = sim.particles
    ps[1].params["min_distance"] = 100.
    sim.post_timestep_modifications(ctypes.byref(sim))
    sim.process_messages()
    assert ps[1].params["min_distance"] == 5.0                   # from particle 0
    ps[1].x = 6.; ps[1].y = 8.
    sim.post_timestep_modifications(ctypes.byref(sim))
    assert ps[1].params["min_distance"] == 5.0                   # further away: kept
    ps[1].x = 0.3; ps[1].y = 0.4
    sim.post_timestep_modifications(ctypes.byref(sim))
    assert ps[1].params["min_distance"] == 0.5
    ps[1].params["min_distance"] = 100.
    ps[1].params["min_distance_from"] = 777
    sim.post_timestep_modifications(ctypes.byref(sim))
    want = ((3. - .3) ** 2 + (4. - .4) ** 2 + 144.) ** 0.5
    assert ps[1].params["min_distance"] == pytest.approx(want, rel=1e-15)
    ps[1].params["min_distance_from"] = 12345                   # nobody carries that hash
    ps[1].params["min_distance"] = 100.
    sim.post_timestep_modifications(ctypes.byref(sim))
    with pytest.raises(RuntimeError, match="not in the simulation"):
        sim.process_messages()
    assert ps[1].params["min_distance"] == 100.
    sim.free_particle_params()


def test_custom_operator_needs_type_and_step_function():
    sim = make_sim()
    rebx = Extras(sim)
    op = rebx.create_operator("mine")
    with pytest.raises(RuntimeError, match="step_function"):
        rebx.add_operator(op, dtfraction=1.)
    calls = []
    op.step_function = lambda s, o, dt: calls.append(dt)
    with pytest.raises(RuntimeError, match="operator_type"):
        rebx.add_operator(op, dtfraction=1.)
    op.operator_type = "recorder"
    rebx.add_operator(op)                                        # recorders: one whole step after
    sim.dt = 0.1
    sim.post_timestep_modifications(ctypes.byref(sim))
    assert calls == [pytest.approx(0.1)]


@pytest.mark.parametrize("integrator", ["euler", "rk2", "rk4", "implicit_midpoint"])
def test_integrate_force_with_a_user_callback_runs_the_schemes_on_the_host(integrator):
    """A linear drag a = -k v has the exact solution v(t) = v0 exp(-k t); each scheme must
    reproduce its own stability polynomial R(-k dt) (no GPU involved: the force is a Python
    callback, so the stages run on the host arrays)."""
    import math
    from reboundx_amd.extras import REBX_INTEGRATORS
    sim = make_sim()
    rebx = Extras(sim)
    k, dt = 0.7, (0.02 if integrator == "implicit_midpoint" else 0.3)   # the fixed-point iteration needs k*dt/2 << 1

    def drag(simp, forcep, particles, n):
        for i in range(n):
            particles[i].ax -= k * particles[i].vx
            particles[i].ay -= k * particles[i].vy
            particles[i].az -= k * particles[i].vz

    force = rebx.create_force("drag")
    force.force_type = "vel"
    force.update_accelerations = drag
    op = rebx.load_operator("integrate_force")
    op.params["force"] = force
    op.params["integrator"] = REBX_INTEGRATORS[integrator]
    sim.particles[1].ax = 123.0                                   # must be cleared first (integrate_force.c:46)
    op.step(sim, dt)
    z = -k * dt
    growth = {"euler": 1 + z, "rk2": 1 + z + z * z / 2, "rk4": 1 + z + z * z / 2 + z ** 3 / 6 + z ** 4 / 24,
              "implicit_midpoint": (1 + z / 2) / (1 - z / 2)}[integrator]
    assert sim.particles[1].vy == pytest.approx(growth, rel=1e-14)
    assert abs(growth - math.exp(z)) < 0.03


def test_integrate_force_without_force_parameter_is_an_error():
    sim = make_sim()
    rebx = Extras(sim)
    op = rebx.load_operator("integrate_force")
    with pytest.raises(RuntimeError, match="Force parameter not set"):
        op.step(sim, 0.1)


def test_host_code_is_clean_under_address_sanitizer():
    """Builds the library's host side with AddressSanitizer + LeakSanitizer (device code as is) with
    tests/native/host_lifecycle.cpp -- attach, parameters, forces, operators and steps, binary
    write/read, truncated files, teardown in REBOUND's order -- and runs it.  No GPU involved."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "asan_check.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host lifecycle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_central_force_acentral_from_a_precession_rate():
    """rebx_central_force_Acentral (src/central_force.c:143-152): G M pomegadot / (1 + gamma/2) / d^(gamma+2) / n with the
    separation and the mean motion of the osculating orbit; gamma = -2 is an error (the precession vanishes) and gives 0.
    The value it returns makes the force do what it says: checked in tests/test_gpu_notebooks.py-style runs is not
    needed here -- the expression is closed-form."""
    import math
    G, M, m, a, e = 2.5, 1.3, 2e-3, 1.7, 0.3
    sim = Simulation(G=G)
    mu = G * (M + m)
    d = a * (1 - e)                                   # planet at pericentre
    v = math.sqrt(mu * (1 + e) / (a * (1 - e)))
    sim.add(m=M, x=0.1, y=-0.2, z=0.05, vx=0.01, vy=0.02, vz=-0.03)
    sim.add(m=m, x=0.1 + d, y=-0.2, z=0.05, vx=0.01, vy=0.02 + v, vz=-0.03)
    rebx = Extras(sim)
    ps = sim.particles
    n = math.sqrt(mu / a ** 3)
    for gamma, rate in ((-1., 1e-3), (-3., -2e-4), (0.5, 7e-6)):
        got = rebx.central_force_Acentral(ps[1], ps[0], rate, gamma)
        want = G * M * rate / (1. + gamma / 2.) / d ** (gamma + 2.) / n
        assert got == pytest.approx(want, rel=1e-13)
    with pytest.raises(RuntimeError, match="Precession vanishes"):
        rebx.central_force_Acentral(ps[1], ps[0], 1e-3, -2.)
    assert lib.rebx_central_force_Acentral(ps[1], ps[0], 1e-3, -2.) == 0.
    with pytest.raises(RuntimeError):
        sim.process_messages()
    # an unbound orbit has a negative mean motion in REBOUND's convention (n = a/|a| sqrt(|mu/a^3|)): the sign carries over
    ps[1].vy = 0.02 + 3. * v
    assert rebx.central_force_Acentral(ps[1], ps[0], 1e-3, -1.) < 0.
    # a primary without mass: REBOUND's orbit is undefined (NaN), and so is the result
    ps[0].m = 0.
    assert math.isnan(rebx.central_force_Acentral(ps[1], ps[0], 1e-3, -1.))
