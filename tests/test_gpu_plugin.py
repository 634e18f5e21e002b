"""The drop-in path on a GPU: host particle arrays go through the plugin API
(rebx_additional_forces / the force callbacks) and must come back with the accelerations the
CPU checker computes -- with parameters living in the reference's linked lists."""
import ctypes

import numpy as np
import pytest

import oracle
from gpu_common import TOL, rel_err, ulp_diff
from reboundx_amd import synthetic
from reboundx_amd.extras import Extras, Particle, Simulation, lib

pytestmark = pytest.mark.gpu
C = synthetic.C_DEFAULT


def sim_from(cols, G=1.0):
    sim = Simulation(G=G)
    n = len(cols["x"])
    for i in range(n):
        sim.add(m=cols["m"][i], x=cols["x"][i], y=cols["y"][i], z=cols["z"][i], vx=cols["vx"][i], vy=cols["vy"][i], vz=cols["vz"][i])
    for i in range(n):
        p = sim.particles[i]
        p.ax, p.ay, p.az = cols["ax"][i], cols["ay"][i], cols["az"][i]
    return sim


def acc_of(particles, n):
    return np.array([[particles[i].ax for i in range(n)], [particles[i].ay for i in range(n)], [particles[i].az for i in range(n)]])


def set_params(sim, pp):
    for key, col in pp.items():
        for i, v in enumerate(col):
            if col.dtype == np.float64:
                if not np.isnan(v):
                    sim.particles[i].params[key] = float(v)
            elif v:
                sim.particles[i].params[key] = 1


def gravity(cols):
    s = oracle.Sim(cols)
    oracle.newtonian_gravity(s)
    cols["ax"], cols["ay"], cols["az"] = [a.copy() for a in s.acc]
    return cols


TWOPLANETS = {
    "radiation_forces": ({"c": 1e4}, {"beta": np.array([np.nan, .5, .3])}),
    "gravitational_harmonics": ({}, {"J2": np.array([1e-3, np.nan, np.nan]), "J4": np.array([1e-3, np.nan, np.nan]), "R_eq": np.array([1e-3, np.nan, np.nan])}),
    "gr": ({"c": 1e4}, {}),
    "gr_potential": ({"c": 1e4}, {}),
    "gr_full": ({"c": 1e4}, {}),
    "tides_precession": ({}, {"R_tides": np.array([1e-3, np.nan, np.nan]), "k1": np.array([.4, np.nan, np.nan])}),
    "central_force": ({}, {"Acentral": np.array([1e-3, np.nan, np.nan]), "gammacentral": np.array([-1., np.nan, np.nan])}),
    "modify_orbits_forces": ({}, {"tau_a": np.array([np.nan, -1e4, np.nan]), "tau_e": np.array([np.nan, -1e3, -1e3]), "tau_inc": np.array([np.nan, np.nan, -1e3])}),
}


@pytest.mark.parametrize("name", sorted(TWOPLANETS))
def test_each_reference_test_setup_through_the_hook(name):
    """The systems and parameters of reboundx/test/test_machine_independent.py, one force
    evaluation through sim->additional_forces, reference-order arithmetic: bitwise equal."""
    ep, pp = TWOPLANETS[name]
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    force = rebx.load_force(name)
    rebx.add_force(force)
    for k, v in ep.items():
        force.params[k] = v
    set_params(sim, pp)
    sim.call_additional_forces()
    got = acc_of(sim.particles, 3)
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [(name, ep)], pp)
    tol_ulps = 16 if name == "central_force" else 0
    assert ulp_diff(got, np.array(ref.acc)).max() <= tol_ulps
    rebx.set_mode("fast")
    for i in range(3):
        sim.particles[i].ax, sim.particles[i].ay, sim.particles[i].az = cols["ax"][i], cols["ay"][i], cols["az"][i]
    sim.call_additional_forces()
    assert rel_err(acc_of(sim.particles, 3), np.array(ref.acc)).max() <= TOL
    sim.free_particle_params()


def test_stack_runs_in_reverse_add_order_fused_and_matches_sequential_reference():
    n = 2000
    cols, pp, _ = synthetic.disk(n, 2)
    synthetic.add_newtonian(cols, n_head=3)
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    added = ["gr", "gravitational_harmonics", "radiation_forces"]          # executed: radiation, harmonics, gr
    for name in added:
        f = rebx.load_force(name)
        if name != "gravitational_harmonics":
            f.params["c"] = C
        rebx.add_force(f)
    set_params(sim, {k: pp[k] for k in ("beta", "J2", "J4", "R_eq")})
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})], pp)
    assert ulp_diff(acc_of(sim.particles, n), np.array(ref.acc)).max() == 0
    t = rebx.last_timings()
    assert t["kernel_ms"] > 0 and t["pack_ms"] > 0
    sim.free_particle_params()


def test_custom_force_between_builtin_forces_sees_and_keeps_the_running_accelerations():
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    seen = {}

    def custom(simp, forcep, particles, N):
        seen["ax1"] = particles[1].ax
        for i in range(N):
            particles[i].ax += 1e-3 * (i + 1)

    gr = rebx.load_force("gr"); gr.params["c"] = 1e4
    mine = rebx.create_force("mine"); mine.update_accelerations = custom; mine.force_type = "pos"
    grp = rebx.load_force("gr_potential"); grp.params["c"] = 1e4
    for f in (gr, mine, grp):                       # executed: gr_potential, mine, gr
        rebx.add_force(f)
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    oracle.gr_potential(ref, 1e4)
    assert seen["ax1"] == ref.cols["ax"][1]
    ref.cols["ax"] += 1e-3 * np.arange(1, 4)
    oracle.gr(ref, 1e4)
    assert ulp_diff(acc_of(sim.particles, 3), np.array(ref.acc)).max() == 0


def test_callback_on_a_scratch_copy_updates_the_copy_only():
    """Split-step integrators hand the callback memcpy'd particle arrays whose parameter
    pointers alias the real ones (integrator_rk4.c:58-83)."""
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    f = rebx.load_force("radiation_forces"); f.params["c"] = 1e4
    sim.particles[1].params["beta"] = 0.5
    scratch = (Particle * 3)()
    ctypes.memmove(scratch, sim.particles, ctypes.sizeof(Particle) * 3)
    scratch[1].x *= 1.01
    before = acc_of(sim.particles, 3)
    lib.rebx_radiation_forces(ctypes.byref(sim), ctypes.byref(f), scratch, 3)
    sim.process_messages()
    assert np.array_equal(acc_of(sim.particles, 3), before)
    c2 = {k: v.copy() for k, v in cols.items()}
    c2["x"][1] *= 1.01
    ref = oracle.Sim(c2)
    oracle.radiation_forces(ref, 1e4, np.array([np.nan, .5, np.nan]), None)
    assert ulp_diff(acc_of(scratch, 3), np.array(ref.acc)).max() == 0
    sim.free_particle_params()


def test_parameter_changes_are_picked_up_also_when_written_through_a_raw_pointer():
    """The reference reads every value from the lists on every call (core.c:601-622), so a caller may keep
    the pointer rebx_get_param returns and overwrite the value: per-particle columns (beta) and source
    parameters (J2) alike must follow without any further call."""
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    f = rebx.load_force("radiation_forces"); f.params["c"] = 1e4
    rebx.add_force(f)
    ps = sim.particles
    ps[1].params["beta"] = 0.5

    def evaluate():
        for i in range(3):
            ps[i].ax, ps[i].ay, ps[i].az = cols["ax"][i], cols["ay"][i], cols["az"][i]
        sim.call_additional_forces()
        return acc_of(ps, 3)

    def expected(b1, b2, J2=None):
        ref = oracle.Sim(cols)
        if J2 is not None:
            oracle.gravitational_harmonics(ref, np.array([J2, np.nan, np.nan]), None, np.array([1e-3, np.nan, np.nan]))
        oracle.radiation_forces(ref, 1e4, np.array([np.nan, b1, b2]), None)
        return np.array(ref.acc)

    assert np.array_equal(evaluate(), expected(0.5, np.nan))
    ps[1].params["beta"] = 0.25                      # update through the API
    assert np.array_equal(evaluate(), expected(0.25, np.nan))
    ps[2].params["beta"] = 0.1                       # new parameter on another particle
    assert np.array_equal(evaluate(), expected(0.25, 0.1))
    beta1 = ctypes.cast(lib.rebx_get_param(ctypes.byref(rebx), ps[1].ap, b"beta"), ctypes.POINTER(ctypes.c_double))
    beta1[0] = 0.75                                  # write through the raw pointer
    assert np.array_equal(evaluate(), expected(0.75, 0.1))
    assert np.array_equal(evaluate(), expected(0.75, 0.1))
    beta1[0] = 0.125
    assert np.array_equal(evaluate(), expected(0.125, 0.1))
    rebx.add_force(rebx.load_force("gravitational_harmonics"))          # runs first (head insertion)
    ps[0].params["J2"] = 1e-3
    ps[0].params["R_eq"] = 1e-3
    assert np.array_equal(evaluate(), expected(0.125, 0.1, 1e-3))
    J2 = ctypes.cast(lib.rebx_get_param(ctypes.byref(rebx), ps[0].ap, b"J2"), ctypes.POINTER(ctypes.c_double))
    J2[0] = 4e-3
    assert np.array_equal(evaluate(), expected(0.125, 0.1, 4e-3))
    sim.free_particle_params()


OTHER_NAN = np.array([0x7FF8000000000001], dtype=np.uint64).view(np.float64)[0]       # a NaN that is not the "not set" marker


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("case", ["beta", "tau_a", "tau_e_with_tau_inc", "tau_e_alone", "R_tides"])
def test_a_parameter_whose_value_is_nan_is_a_parameter_and_poisons_what_the_reference_poisons(case, mode):
    """A parameter set to NaN is found in the list and computed with (core.c:614-622; e.g.
    modify_orbits_forces.c:94-108): the accelerations it enters become NaN -- its own particle's, and the
    source's through the back-reaction -- and nothing else does.  (tau_e alone: the reference's test
    `tau_e < INFINITY || tau_inc < INFINITY`, modify_orbits_forces.c:110, is false for NaN, so the term is
    skipped and nothing is poisoned.)  The checker takes the same columns with the value NaN marked as a value."""
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode(mode)
    nan = float("nan")
    ps = sim.particles
    un = np.nan
    if case == "beta":
        f = rebx.load_force("radiation_forces"); f.params["c"] = 1e4
        ps[1].params["beta"] = nan; ps[2].params["beta"] = 0.3
        stack, pp = [("radiation_forces", {"c": 1e4})], {"beta": np.array([un, OTHER_NAN, 0.3])}
    elif case == "R_tides":
        f = rebx.load_force("tides_precession")
        ps[0].params["R_tides"] = 1e-3; ps[0].params["k1"] = 0.4
        ps[1].params["R_tides"] = nan; ps[1].params["k1"] = 0.3
        stack, pp = [("tides_precession", {})], {"R_tides": np.array([1e-3, OTHER_NAN, un]), "k1": np.array([.4, .3, un])}
    else:
        f = rebx.load_force("modify_orbits_forces"); f.params["coordinates"] = 2
        ps[0].params["primary"] = 1
        ps[2].params["tau_a"] = -1e4
        pp = {"tau_a": np.array([un, un, -1e4]), "tau_e": np.array([un, un, un]), "tau_inc": np.array([un, un, un]), "primary": np.array([1, 0, 0], dtype=np.uint8)}
        if case == "tau_a":
            ps[1].params["tau_a"] = nan; pp["tau_a"][1] = OTHER_NAN
        else:
            ps[1].params["tau_e"] = nan; pp["tau_e"][1] = OTHER_NAN
            if case == "tau_e_with_tau_inc":
                ps[1].params["tau_inc"] = -1e3; pp["tau_inc"][1] = -1e3
        stack = [("modify_orbits_forces", {"coordinates": 2})]
    rebx.add_force(f)
    sim.call_additional_forces()
    got = acc_of(ps, 3)
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, stack, pp)
    want = np.array(ref.acc)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    assert np.isnan(want).any() == (case != "tau_e_alone")
    ok = ~np.isnan(want)
    if mode == "exact":
        assert np.array_equal(got[ok], want[ok])
    else:
        assert (np.abs(got[ok] - want[ok]) <= TOL * np.abs(want[ok])).all()
    sim.free_particle_params()


def big_array_sim(cols):
    n = len(cols["x"])
    sim = Simulation()
    arr = (Particle * n)()
    view = np.frombuffer(arr, dtype=np.float64).reshape(n, 16)
    for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "m")):
        view[:, j] = cols[k]
    sim._store = arr
    sim._particles = ctypes.cast(arr, ctypes.POINTER(Particle))
    sim.N = n
    return sim, arr, view


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_parameter_columns_equal_one_hundred_thousand_per_particle_setters_bit_for_bit(mode):
    """rebxamd_set_param_column against rebx_set_param_double on every particle (core.c:552-565), N = 1e5,
    radiation + tides + drag stacked, through sim->additional_forces: the same accelerations to the bit.
    Then the precedence rules: a particle's own list wins over its column entry, a column shorter than the
    array leaves the rest not set, and removing the column removes the force's effect."""
    n = 100_000
    cols, pp, _ = synthetic.disk(n, 2, seed=5)
    cols["m"][3:] = 1e-12                           # tides and the drag's back-reaction act on massive bodies
    names = ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc")
    rng = np.random.default_rng(11)
    pp = {k: pp[k].copy() for k in names}
    pp["R_tides"][3:] = 1e-5; pp["k1"][3:] = 0.3
    pp["R_tides"][0] = 1e-3; pp["k1"][0] = 0.4
    for k in names:                                 # holes everywhere: a third of the bodies lack each parameter
        holes = rng.random(n) < 0.33
        holes[0] = k not in ("R_tides", "k1")
        pp[k][holes] = np.nan

    def build(with_columns):
        sim, arr, view = big_array_sim(cols)
        rebx = Extras(sim)
        rebx.set_mode(mode)
        for name in ("modify_orbits_forces", "tides_precession", "radiation_forces"):
            f = rebx.load_force(name)
            if name == "radiation_forces":
                f.params["c"] = C
            if name == "modify_orbits_forces":
                f.params["coordinates"] = 2
            rebx.add_force(f)
        arr[0]._sim = ctypes.pointer(sim)
        arr[0].params["primary"] = 1
        if with_columns:
            for k in names:
                assert rebx.set_param_column(k, pp[k])
        else:
            ap = ctypes.c_void_p.from_buffer
            for k in names:
                key = k.encode("ascii")
                for i in np.nonzero(~np.isnan(pp[k]))[0]:
                    lib.rebx_set_param_double(ctypes.byref(rebx), ctypes.byref(ap(arr[i], Particle.ap.offset)), key, float(pp[k][i]))
        return sim, arr, view, rebx

    sim_l, arr_l, view_l, rebx_l = build(False)
    sim_l.call_additional_forces()
    want = view_l[:, 6:9].copy()
    sim_c, arr_c, view_c, rebx_c = build(True)
    sim_c.call_additional_forces()
    got = view_c[:, 6:9].copy()
    assert np.isfinite(want).all() and np.abs(want - np.array([cols["ax"], cols["ay"], cols["az"]]).T).max() > 0
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    some = int(np.nonzero(~np.isnan(pp["beta"]))[0][5])
    assert not arr_c[some].ap and arr_l[some].ap          # columns make no list nodes

    ref = oracle.Sim(cols)                          # and both are the checker's answer
    pp_ref = dict(pp); pp_ref["primary"] = np.zeros(n, dtype=np.uint8); pp_ref["primary"][0] = 1
    oracle.apply_forces(ref, [("radiation_forces", {"c": C}), ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2})], pp_ref)
    refacc = np.array(ref.acc).T
    if mode == "exact":
        assert ulp_diff(got[1:], refacc[1:]).max() == 0
    assert rel_err(got.T, refacc.T).max() <= TOL

    def again(sim, view):
        for j, k in enumerate(("ax", "ay", "az")):
            view[:, 6 + j] = cols[k]
        sim.call_additional_forces()
        return view[:, 6:9].copy()

    # a list entry wins over the column; the list-only simulation with the same value agrees
    i = 4242
    arr_c[i]._sim = ctypes.pointer(sim_c); arr_l[i]._sim = ctypes.pointer(sim_l)
    arr_c[i].params["tau_a"] = -77.; arr_l[i].params["tau_a"] = -77.
    a_c, a_l = again(sim_c, view_c), again(sim_l, view_l)
    assert np.array_equal(a_c.view(np.uint64), a_l.view(np.uint64))
    assert not np.array_equal(a_c[i], got[i])
    # a shorter column: bodies behind its end have the parameter not set
    half = n // 2
    assert rebx_c.set_param_column("beta", pp["beta"][:half])
    pp_half = dict(pp_ref); pp_half["beta"] = pp["beta"].copy(); pp_half["beta"][half:] = np.nan
    pp_half["tau_a"] = pp["tau_a"].copy(); pp_half["tau_a"][i] = -77.
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [("radiation_forces", {"c": C}), ("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2})], pp_half)
    assert rel_err(again(sim_c, view_c).T, np.array(ref.acc)).max() <= TOL
    # removing the columns: only the list entry is left
    for k in names:
        assert rebx_c.set_param_column(k, None)
    left = again(sim_c, view_c)
    moved = np.nonzero((left != np.array([cols["ax"], cols["ay"], cols["az"]]).T).any(axis=1))[0]
    assert set(moved) == {0, i}
    sim_c.free_particle_params(); sim_l.free_particle_params()


def test_missing_required_parameters_report_errors_and_leave_accelerations_alone():
    cols = gravity(synthetic.twoplanets())
    for name, match in (("radiation_forces", "speed of light"), ("gr", "speed of light"), ("gr_full", "speed of light"),
                        ("gr_potential", "speed of light")):
        sim = sim_from(cols)
        rebx = Extras(sim)
        rebx.add_force(rebx.load_force(name))
        with pytest.raises(RuntimeError, match=match):
            sim.call_additional_forces()
        assert np.array_equal(acc_of(sim.particles, 3), np.array([cols["ax"], cols["ay"], cols["az"]]))
    sim = sim_from(cols)
    rebx = Extras(sim)
    mof = rebx.load_force("modify_orbits_forces")
    mof.params["coordinates"] = 2
    rebx.add_force(mof)
    with pytest.raises(RuntimeError, match="primary param was not found"):
        sim.call_additional_forces()


def test_n_var_particles_are_excluded():
    cols = gravity(synthetic.twoplanets())
    sim = sim_from(cols)
    sim.N_var = 1                                    # last particle is a variational particle
    rebx = Extras(sim)
    rebx.set_mode("exact")
    f = rebx.load_force("gr_potential"); f.params["c"] = 1e4
    rebx.add_force(f)
    sim.call_additional_forces()
    two = {k: v[:2].copy() for k, v in cols.items()}
    ref = oracle.Sim(two)
    oracle.gr_potential(ref, 1e4)
    got = acc_of(sim.particles, 3)
    assert ulp_diff(got[:, :2], np.array(ref.acc)).max() == 0
    assert np.array_equal(got[:, 2], np.array([cols["ax"][2], cols["ay"][2], cols["az"][2]]))


def test_larger_host_array_round_trip_and_timings():
    n = 200_000
    cols, pp, _ = synthetic.disk(n, 1)
    synthetic.add_newtonian(cols, n_head=2)
    sim = Simulation()
    arr = (Particle * n)()
    view = np.frombuffer(arr, dtype=np.float64).reshape(n, 16)
    for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "m")):
        view[:, j] = cols[k]
    sim._store = arr
    sim._particles = ctypes.cast(arr, ctypes.POINTER(Particle))
    sim.N = n
    for i in (0, 1):
        arr[i]._sim = ctypes.pointer(sim)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    f = rebx.load_force("gr"); f.params["c"] = C
    rebx.add_force(f)
    g = rebx.load_force("gravitational_harmonics")
    rebx.add_force(g)
    arr[0].params["J2"] = 1e-3
    arr[0].params["R_eq"] = 1e-3
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    J2 = np.full(n, np.nan); J2[0] = 1e-3
    oracle.apply_forces(ref, [("gravitational_harmonics", {}), ("gr", {"c": C})], {"J2": J2, "R_eq": J2})
    got = np.array([view[:, 6], view[:, 7], view[:, 8]])
    assert ulp_diff(got, np.array(ref.acc)).max() == 0
    sim.free_particle_params()


def test_large_array_in_fast_mode_adds_onto_host_accelerations_and_follows_mass_changes():
    """Fast mode on a large array: the accelerations are not uploaded (the device sums from zero, the host adds) and a
    block of masses travels only when it changed.  Incoming accelerations (gravity here) must survive, and a changed
    star mass, a changed array length and a changed mass far down the array must each show in the next call."""
    n = 1_200_000                                   # more than one block of the upload pipeline
    cols, pp, _ = synthetic.disk(n, 1)
    synthetic.add_newtonian(cols, n_head=1)
    sim, arr, view = big_array_sim(cols)
    rebx = Extras(sim)
    f = rebx.load_force("radiation_forces"); f.params["c"] = C
    rebx.add_force(f)
    g = rebx.load_force("gr_potential"); g.params["c"] = C
    rebx.add_force(g)
    arr[0]._sim = ctypes.pointer(sim)
    arr[0].params["radiation_source"] = 1
    rebx.set_param_column("beta", pp["beta"])
    src = np.zeros(n, dtype=np.uint8); src[0] = 1

    def check(count):
        for j, k in enumerate(("ax", "ay", "az")):      # what REBOUND does before every evaluation: gravity, afresh
            view[:, 6 + j] = cols[k]
        untouched = view[:, 6:9].copy()
        start = {k: view[:count, 6 + j].copy() for j, k in enumerate(("ax", "ay", "az"))}
        now = {k: view[:count, j].copy() for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz"))}
        now.update(start); now["m"] = view[:count, 9].copy()
        sim.N = count
        sim.call_additional_forces()
        ref = oracle.Sim(now)
        oracle.apply_forces(ref, [("gr_potential", {"c": C}), ("radiation_forces", {"c": C})],
                            {"beta": pp["beta"][:count], "radiation_source": src[:count]})
        got, want = np.array([view[:count, 6], view[:count, 7], view[:count, 8]]), np.array(ref.acc)
        # 1e-12 of the body's total or of what was added to it, whichever is larger (radiation pressure cancels
        # most of gravity where beta is near one: the sum then amplifies the rounding of both)
        before = np.array([start["ax"], start["ay"], start["az"]])
        scale = np.maximum(np.abs(want).max(axis=0), np.abs(want - before).max(axis=0))
        assert (np.abs(got - want) / scale[None, :]).max() <= TOL
        assert np.array_equal(view[count:, 6:9], untouched[count:])

    check(n)
    view[0, 9] *= 1.5                               # the star's mass: first block only
    check(n)
    check(n - 100_000)                              # a shorter array ...
    view[n - 5, 9] = 2e-9                           # ... and a longer one again whose new end holds a changed mass
    check(n)
    sim.free_particle_params()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_position_only_forces_on_a_large_array_never_look_at_the_velocities(mode):
    """A batch of forces the reference types REBX_FORCE_POS (core.c:232-263) uploads no velocities: the result must be the
    checker's whatever the host velocities hold (NaN here), on a fresh device buffer."""
    n = 300_000
    cols, pp, _ = synthetic.disk(n, 1)
    synthetic.add_newtonian(cols, n_head=1)
    sim, arr, view = big_array_sim(cols)
    view[:, 3:6] = np.nan
    rebx = Extras(sim)
    rebx.set_mode(mode)
    g = rebx.load_force("gr_potential"); g.params["c"] = C
    rebx.add_force(g)
    rebx.add_force(rebx.load_force("gravitational_harmonics"))
    arr[0]._sim = ctypes.pointer(sim)
    arr[0].params["J2"] = 1e-3; arr[0].params["J4"] = -3e-5; arr[0].params["R_eq"] = 2e-3
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    col = lambda v: np.concatenate([[v], np.full(n - 1, np.nan)])
    oracle.apply_forces(ref, [("gravitational_harmonics", {}), ("gr_potential", {"c": C})], {"J2": col(1e-3), "J4": col(-3e-5), "R_eq": col(2e-3)})
    got, want = np.array([view[:, 6], view[:, 7], view[:, 8]]), np.array(ref.acc)
    assert np.isfinite(got).all()
    if mode == "exact":
        assert ulp_diff(got[:, 1:], want[:, 1:]).max() == 0
    assert rel_err(got, want).max() <= TOL
    sim.free_particle_params()


def test_parameter_lists_spread_over_a_large_array_are_flattened_by_several_host_threads():
    """Above 1e5 particles the host side converts arrays and walks the parameter lists on several
    threads, each over its own range of particles.  Parameters set on particles of every range, a
    second radiation source and a second oblate body far from the first (found by different
    threads, merged in index order) must give what the checker computes from the same columns."""
    n = 160_000
    cols, pp, _ = synthetic.disk(n, 1)
    far = 150_001
    cols["m"][far] = 3e-4
    sim = Simulation()
    arr = (Particle * n)()
    view = np.frombuffer(arr, dtype=np.float64).reshape(n, 16)
    for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "m")):
        view[:, j] = cols[k]
    sim._store = arr
    sim._particles = ctypes.cast(arr, ctypes.POINTER(Particle))
    sim.N = n
    rebx = Extras(sim)
    rebx.set_mode("exact")
    f = rebx.load_force("radiation_forces"); f.params["c"] = C
    rebx.add_force(f)
    rebx.add_force(rebx.load_force("gravitational_harmonics"))
    chosen = np.arange(5, n, 131)
    beta = np.full(n, np.nan); beta[chosen] = pp["beta"][chosen]
    J2 = np.full(n, np.nan); J4 = np.full(n, np.nan); Req = np.full(n, np.nan)
    src = np.zeros(n, dtype=np.uint8)
    for i in (0, far):
        arr[i]._sim = ctypes.pointer(sim)
        arr[i].params["radiation_source"] = 1
        arr[i].params["J2"] = 1e-3 * (1 + i % 7); arr[i].params["R_eq"] = 1e-3
        J2[i], Req[i], src[i] = 1e-3 * (1 + i % 7), 1e-3, 1
    arr[far].params["J4"] = -2e-4; J4[far] = -2e-4
    for i in chosen:
        arr[i]._sim = ctypes.pointer(sim)
        arr[i].params["beta"] = float(beta[i])
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [("gravitational_harmonics", {}), ("radiation_forces", {"c": C})],
                        {"beta": beta, "J2": J2, "J4": J4, "R_eq": Req, "radiation_source": src})
    got = np.array([view[:, 6], view[:, 7], view[:, 8]])
    want = np.array(ref.acc)
    # the oblate bodies collect the pushes of every body with mass as parallel sums
    massive = np.nonzero(cols["m"])[0]
    rest = np.setdiff1d(np.arange(n), massive)
    assert ulp_diff(got[:, rest], want[:, rest]).max() == 0
    assert rel_err(got[:, massive], want[:, massive]).max() <= TOL
    sim.free_particle_params()


def test_gr_and_jacobi_drag_with_massive_bodies_anywhere_through_the_hook():
    """Planets behind massless particles (no massive prefix): the hook must still run gr and
    Jacobi-coordinate modify_orbits_forces on the device (prefix-sum path).  Tolerance: 1e-12
    relative to the largest component of the body's update (sequential vs. parallel order of the
    interior-mass sums; see test_gpu_parity.scan_check)."""
    n = 600
    cols, pp, _ = synthetic.disk(n, 3)
    perm = np.concatenate([[0], 1 + np.random.default_rng(2).permutation(n - 1)])
    cols = {k: np.ascontiguousarray(v[perm]) for k, v in cols.items()}
    pp = {k: np.ascontiguousarray(v[perm]) for k, v in pp.items()}
    sim = sim_from(cols)
    rebx = Extras(sim)
    for name in ("modify_orbits_forces", "gr"):               # executed: gr, then mof (Jacobi by default)
        f = rebx.load_force(name)
        if name == "gr":
            f.params["c"] = C
        rebx.add_force(f)
    set_params(sim, {k: pp[k] for k in ("tau_a", "tau_e", "tau_inc")})
    sim.call_additional_forces()
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, [("gr", {"c": C}), ("modify_orbits_forces", {})], pp)
    got, want = acc_of(sim.particles, n), np.array(ref.acc)
    big = np.abs(want).max(axis=0)
    assert (np.abs(got - want) / big[None, :]).max() <= TOL
    sim.free_particle_params()


def test_parameter_columns_feed_the_operators_too():
    """tau_mass (modify_mass) and tau_a / tau_e / tau_omega (modify_orbits_direct) given as columns: the operators' step
    functions must do what they do with the same values in the particles' lists -- and the same through one copy per
    direction as evaluated in the pinned host columns (the default at this size)."""
    n = 3000
    cols, pp, _ = synthetic.disk(n, 2)
    rng = np.random.default_rng(4)
    tau_mass = np.where(rng.random(n) < 0.5, -1e3 * (1 + rng.random(n)), np.nan)
    tau_omega = np.where(rng.random(n) < 0.3, 100. * (1 + rng.random(n)), np.nan)
    given = {"tau_mass": tau_mass, "tau_a": pp["tau_a"], "tau_e": pp["tau_e"], "tau_omega": tau_omega}
    states = []
    for how in ("lists", "columns", "columns, copied"):
        sim, arr, view = big_array_sim(cols)
        sim.integrator = 1
        sim.dt = 0.1
        rebx = Extras(sim)
        rebx.set_in_place(how != "columns, copied")
        mm = rebx.load_operator("modify_mass")
        mod = rebx.load_operator("modify_orbits_direct")
        if how != "lists":
            for k, v in given.items():
                assert rebx.set_param_column(k, v)
        else:
            for k, v in given.items():
                key = k.encode("ascii")
                for i in np.nonzero(~np.isnan(v))[0]:
                    lib.rebx_set_param_double(ctypes.byref(rebx), ctypes.byref(ctypes.c_void_p.from_buffer(arr[i], Particle.ap.offset)), key, float(v[i]))
        mm.step(sim, 0.05)
        mod.step(sim, 0.05)
        sim.process_messages()
        states.append(view[:, [0, 1, 2, 3, 4, 5, 9]].copy())
        sim.free_particle_params()
    start = np.array([cols[k] for k in ("x", "y", "z", "vx", "vy", "vz", "m")]).T
    assert np.abs(states[0] - start).max() > 1e-7
    assert np.array_equal(states[0].view(np.uint64), states[1].view(np.uint64))
    assert np.array_equal(states[1].view(np.uint64), states[2].view(np.uint64))


@pytest.mark.parametrize("in_place", [True, False], ids=["in-place", "copies"])
@pytest.mark.parametrize("integrator", ["euler", "rk2", "rk4", "implicit_midpoint"])
def test_integrate_force_operator_with_a_builtin_force_runs_on_the_device(integrator, in_place):
    """The reference's IntegrateForce setup (ipython_examples/IntegrateForce.ipynb: a
    velocity-dependent force applied as an operator): built-in force + integrate_force operator,
    stepped once through the operator's step function.  Reference-order arithmetic: the explicit
    schemes are bitwise equal to the checker's restatement -- evaluated in the pinned host columns
    (the default for a system this small) and through one copy per direction alike."""
    from reboundx_amd.extras import REBX_INTEGRATORS
    n = 1500
    cols, pp, _ = synthetic.disk(n, 2)
    synthetic.add_newtonian(cols, n_head=3)          # whatever is in the accelerations must be cleared first
    sim = sim_from(cols)
    rebx = Extras(sim)
    rebx.set_mode("exact")
    rebx.set_in_place(in_place)
    force = rebx.load_force("modify_orbits_forces")
    op = rebx.load_operator("integrate_force")
    op.params["force"] = force
    op.params["integrator"] = REBX_INTEGRATORS[integrator]
    set_params(sim, {k: pp[k] for k in ("tau_a", "tau_e", "tau_inc")})
    dt = 0.05
    op.step(sim, dt)
    ref = oracle.Sim(cols)
    oracle.integrate_force(ref, integrator, dt, [("modify_orbits_forces", {})], pp)
    got = np.array([[sim.particles[i].vx for i in range(n)], [sim.particles[i].vy for i in range(n)], [sim.particles[i].vz for i in range(n)]])
    want = np.array([ref.cols[k] for k in ("vx", "vy", "vz")])
    if integrator == "implicit_midpoint":
        dv = np.abs(want - np.array([cols[k] for k in ("vx", "vy", "vz")])).max(axis=0)
        assert (np.abs(got - want).max(axis=0) <= TOL * np.maximum(dv, 1e-3)).all()
    else:
        assert ulp_diff(got, want).max() == 0
        assert ulp_diff(acc_of(sim.particles, n), np.array(ref.acc)).max() == 0
    sim.free_particle_params()


@pytest.mark.parametrize("name", ["modify_mass", "modify_orbits_direct"])
def test_operator_fixture_of_the_reference_runs_through_the_timestep_hooks(name):
    """Load the reference's own .rebx fixture (operator + steps + particle timescales) into its
    three-body system and let the pre-timestep hook run the operator for half a step on the
    device; compare with the checker's operator, which reproduces the reference's archive."""
    import json
    import os
    import warnings as _w
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    arch = json.load(open(os.path.join(golden, "reference_archives.json")))
    start = np.array([[float.fromhex(v) for v in row] for row in arch["forces"][name][0]["particles"]])
    cols = {k: start[:, j].copy() for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "m"))}
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(3)
    sim = sim_from(cols)
    sim.integrator = 1                       # whfast
    sim.dt = float.fromhex(arch["dt"])
    with _w.catch_warnings():
        _w.simplefilter("ignore")            # written by an older version
        rebx = Extras(sim, os.path.join(golden, "rebx", name + ".rebx"))
    sim.pre_timestep_modifications(ctypes.byref(sim))
    sim.process_messages()
    ref = oracle.Sim(cols)
    nan = lambda **kw: np.array([kw.get(i, np.nan) for i in (0, 1, 2)])
    if name == "modify_mass":
        oracle.modify_mass(ref, 0.5 * sim.dt, np.array([-1e4, np.nan, np.nan]))
    else:
        oracle.modify_orbits_direct(ref, 0.5 * sim.dt, 0, None, np.array([np.nan, -1e4, np.nan]), np.array([np.nan, -1e3, -1e3]),
                                    np.array([np.nan, np.nan, -1e3]), np.array([np.nan, 100., np.nan]), None, None)
    got = np.array([[getattr(sim.particles[i], k) for i in range(3)] for k in ("x", "y", "z", "vx", "vy", "vz", "m")])
    want = np.array([ref.cols[k] for k in ("x", "y", "z", "vx", "vy", "vz", "m")])
    assert np.abs(got - want).max() <= 1e-13
    assert np.abs(want - start.T).max() > 1e-7
    sim.free_particle_params()
    del rebx


def test_two_simulations_stepping_on_two_host_threads_share_the_helper_threads():
    """The array-of-structs <-> column conversions run on helper threads that are started once and parked between calls; they
    take one job at a time, and a caller that finds them busy starts threads of its own.  Two simulations of 3e5 particles
    evaluated from two host threads at once (ctypes releases the interpreter lock inside the call), several times over, must
    each get what they get alone -- bit for bit in the reference-order arithmetic -- and a run of calls on one thread after
    that must still find the helpers alive."""
    import threading
    n = 300_000
    sims = []
    for seed_shift in (0, 1):
        cols, pp, _ = synthetic.disk(n, 1)
        cols["x"][2:] += 1e-3 * seed_shift                     # two different systems
        synthetic.add_newtonian(cols, n_head=2)
        sim, arr, view = big_array_sim(cols)
        for i in (0, 1):
            arr[i]._sim = ctypes.pointer(sim)
        rebx = Extras(sim)
        rebx.set_mode("exact")
        f = rebx.load_force("gr"); f.params["c"] = C
        rebx.add_force(f)
        sims.append((sim, rebx, view, np.array([cols["ax"], cols["ay"], cols["az"]]).T.copy()))

    def evaluate(entry, times, out):
        sim, _, view, start = entry
        for _ in range(times):
            view[:, 6:9] = start
            sim.call_additional_forces()
        out.append(view[:, 6:9].copy())

    alone = []
    for entry in sims:
        got = []
        evaluate(entry, 1, got)
        alone.append(got[0])
    assert not np.array_equal(alone[0], alone[1])
    together = [[], []]
    threads = [threading.Thread(target=evaluate, args=(sims[k], 6, together[k])) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in (0, 1):
        assert np.array_equal(together[k][0], alone[k])
    again = []
    evaluate(sims[0], 3, again)
    assert np.array_equal(again[0], alone[0])


def debris_disk_example():
    """examples/rad_forces_debris_disk/problem.c:21-74 with its literal numbers: SI units (G = 6.674e-11), a star of
    1.99e30 kg, 1000 massless grains on coplanar orbits with a in [100, 120] AU (AU = 1.5e11 m), e = 0.1, pericentre and
    true anomaly drawn with the C library's rand() after srand(3) in the example's order, then the shift to the centre of
    mass; radiation_forces with c = 3e8 and beta = 0.1 on every grain.  The orbit -> Cartesian step restates REBOUND's
    reb_tools_orbit2d_to_particle for a planar orbit about a star at rest at the origin."""
    libc = ctypes.CDLL(None)
    libc.rand.restype = ctypes.c_int
    rand_max = 2147483647.0
    AU, G, M, e, ndust = 1.5e11, 6.674e-11, 1.99e30, 0.1, 1000
    amin, awidth = 100. * AU, 20. * AU
    libc.srand(3)
    rows = [(0., 0., 0., 0., 0., 0., M)]
    for _ in range(ndust):
        a = amin + awidth * libc.rand() / rand_max
        pomega = 2 * np.pi * libc.rand() / rand_max
        f = 2 * np.pi * libc.rand() / rand_max
        r = a * (1 - e * e) / (1 + e * np.cos(f))
        v0 = np.sqrt(G * M / a / (1. - e * e))
        rows.append((r * np.cos(pomega + f), r * np.sin(pomega + f), 0.,
                     v0 * (-np.sin(pomega + f) - e * np.sin(pomega)), v0 * (np.cos(pomega + f) + e * np.cos(pomega)), 0., 0.))
    p = np.array(rows)
    cols = {k: p[:, j].copy() for j, k in enumerate(("x", "y", "z", "vx", "vy", "vz", "m"))}
    # reb_move_to_com: only the star has mass, and it sits at the origin at rest -- the shift is exactly zero
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(ndust + 1)
    beta = np.full(ndust + 1, np.nan)
    beta[1:] = 0.1
    return cols, G, beta


@pytest.mark.parametrize("incoming", ["zero", "gravity"])
def test_debris_disk_example_with_its_literal_si_parameters(incoming):
    """BASELINE config 1 as the reference's example states it -- squared distances of 2e26 m^2, accelerations of 7e-8 m/s^2,
    velocities of 3e3 m/s against c = 3e8 -- through the sim->additional_forces hook with the parameters set one by one
    (1000 x rebx_set_param_double), and through rebxamd_apply on columns; both arithmetic modes against the checker:
    0 ulp in reference order, 1e-12 per component in the default arithmetic."""
    from gpu_common import run_device
    from reboundx_amd.device import Context
    cols, G, beta = debris_disk_example()
    n = len(beta)
    assert 1.8e26 < (cols["x"][1:] ** 2 + cols["y"][1:] ** 2).min() and (cols["x"][1:] ** 2 + cols["y"][1:] ** 2).max() < 4e26
    if incoming == "gravity":
        s = oracle.Sim(cols, G)
        oracle.newtonian_gravity(s)
        cols["ax"], cols["ay"], cols["az"] = [a.copy() for a in s.acc]
    stack, pp = [("radiation_forces", {"c": 3.e8})], {"beta": beta}
    ref = oracle.Sim(cols, G)
    oracle.apply_forces(ref, stack, pp)
    want = np.array(ref.acc)
    base = np.array([cols["ax"], cols["ay"], cols["az"]])
    assert 1e-8 < np.abs(want - base)[:, 1:].max() < 1e-6              # beta G M / r^2 ~ 7e-8 m/s^2

    sim = sim_from(cols, G)
    sim.integrator = 1                                                  # WHFast, as in the example
    sim.N_active = 1
    rebx = Extras(sim)
    rad = rebx.load_force("radiation_forces")
    rebx.add_force(rad)
    rad.params["c"] = 3.e8
    ps = sim.particles
    for i in range(1, n):
        ps[i].params["beta"] = 0.1
    for mode in ("exact", "fast"):
        rebx.set_mode(mode)
        for i in range(n):
            ps[i].ax, ps[i].ay, ps[i].az = cols["ax"][i], cols["ay"][i], cols["az"][i]
        sim.call_additional_forces()
        got = acc_of(ps, n)
        if mode == "exact":
            assert ulp_diff(got, want).max() == 0
        else:
            # the default arithmetic leaves the accelerations on the host and adds the summed forces once
            assert rel_err(got - base, want - base).max() <= TOL
            assert rel_err(got, want).max() <= TOL
    sim.free_particle_params()

    ctx = Context(0)
    try:
        exact, _ = run_device(ctx, cols, pp, stack, G, 1)
        fast, _ = run_device(ctx, cols, pp, stack, G, 0)
    finally:
        ctx.close()
    assert ulp_diff(exact, want).max() == 0
    assert rel_err(fast - base, want - base).max() <= TOL and rel_err(fast, want).max() <= TOL


def test_reference_order_arithmetic_through_the_hook_on_a_scattered_mass_layout():
    """Massive bodies anywhere in the array (no massless tail behind a short head): in reference-order arithmetic the
    hook walks gr and Jacobi / barycentric modify_orbits_forces in the reference's own order, so every particle comes
    back with the checker's bits; the default arithmetic (parallel prefix sums) stays within 1e-12 of each particle's
    update."""
    n = 1500
    cols, pp, _ = synthetic.disk(n, 3)
    rng = np.random.default_rng(11)
    for i in rng.choice(np.arange(4, n), size=12, replace=False):
        cols["m"][i] = 10.0 ** rng.uniform(-7, -4)
    cols = gravity(cols)
    sim = sim_from(cols)
    rebx = Extras(sim)
    added = [("modify_orbits_forces", {"coordinates": 1}), ("gr", {"c": C}), ("modify_orbits_forces", {})]      # executed in reverse
    for name, ep in added:
        f = rebx.load_force(name)
        for k, v in ep.items():
            f.params[k] = v
        rebx.add_force(f)
    taus = {k: pp[k] for k in ("tau_a", "tau_e", "tau_inc")}
    set_params(sim, taus)
    ref = oracle.Sim(cols)
    oracle.apply_forces(ref, list(reversed(added)), taus)
    want = np.array(ref.acc)
    base = np.array([cols["ax"], cols["ay"], cols["az"]])
    ps = sim.particles
    for mode in ("exact", "fast"):
        rebx.set_mode(mode)
        for i in range(n):
            ps[i].ax, ps[i].ay, ps[i].az = cols["ax"][i], cols["ay"][i], cols["az"][i]
        sim.call_additional_forces()
        got = acc_of(ps, n)
        if mode == "exact":
            assert ulp_diff(got, want).max() == 0
        else:
            upd = np.abs(want - base).max(axis=0)
            assert (np.abs(got - want).max(axis=0) <= TOL * np.maximum(upd, np.abs(want).max(axis=0) * 1e-3)).all()
    sim.free_particle_params()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_small_systems_evaluated_in_the_pinned_host_columns_equal_the_copied_route_bit_for_bit(mode):
    """The hook evaluates a small system where it packs it (kernels reading and updating pinned host memory over the
    link) or, switched off, through one copy per direction: the same kernels on the same numbers, so the same bits --
    over several calls in a row with the accelerations carried along, after a parameter and a position change, and
    when a grain beyond the 64-body head gains mass (the route then falls back to the copies by itself)."""
    n = 3000
    cols, pp, _ = synthetic.disk(n, 2)
    cols = gravity(cols)

    def build():
        sim = sim_from(cols)
        rebx = Extras(sim)
        rebx.set_mode(mode)
        for name, ep in (("radiation_forces", {"c": C}), ("gr", {"c": C}), ("gravitational_harmonics", {}), ("modify_orbits_forces", {})):
            f = rebx.load_force(name)
            for k, v in ep.items():
                f.params[k] = v
            rebx.add_force(f)
        ps = sim.particles
        ps[0].params["J2"], ps[0].params["J4"], ps[0].params["R_eq"] = 1e-3, -2e-5, 2e-3
        ps[0].params["radiation_source"] = 1
        rebx.set_param_column("beta", pp["beta"])
        rebx.set_param_column("tau_a", pp["tau_a"])
        rebx.set_param_column("tau_e", pp["tau_e"])
        return sim, rebx

    (sim_a, rebx_a), (sim_b, rebx_b) = build(), build()
    rebx_b.set_in_place(False)
    want = oracle.Sim(cols)
    stack = [("modify_orbits_forces", {}), ("gravitational_harmonics", {}), ("gr", {"c": C}), ("radiation_forces", {"c": C})]      # executed in reverse add order
    par = {"beta": pp["beta"], "tau_a": pp["tau_a"], "tau_e": pp["tau_e"], "J2": pp["J2"] * 0 + np.nan, "J4": pp["J4"] * 0 + np.nan, "R_eq": pp["R_eq"] * 0 + np.nan}
    par["J2"][0], par["J4"][0], par["R_eq"][0] = 1e-3, -2e-5, 2e-3
    oracle.apply_forces(want, stack, par)
    for step in range(5):
        if step == 2:
            for sim in (sim_a, sim_b):
                sim.particles[7].x *= 1.01
                sim.particles[0].params["J2"] = 2e-3
        if step == 4:
            for sim in (sim_a, sim_b):
                sim.particles[n - 5].m = 1e-9
        sim_a.call_additional_forces()
        sim_b.call_additional_forces()
        got_a, got_b = acc_of(sim_a.particles, n), acc_of(sim_b.particles, n)
        assert np.isfinite(got_a).all()
        assert ulp_diff(got_a, got_b).max() == 0, step
        if step == 0:
            if mode == "exact":
                assert ulp_diff(got_a, np.array(want.acc)).max() == 0
            else:
                assert rel_err(got_a, np.array(want.acc)).max() <= TOL
    for sim in (sim_a, sim_b):
        sim.free_particle_params()
