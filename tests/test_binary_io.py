"""Reading and writing the reference's .rebx parameter files (no GPU needed).

Fixtures: tests/golden/rebx/*.rebx are the files the reference's own test suite holds
(reboundx/test/binaries/*.rebx, written by REBOUNDx 2.19.3 with the parameters listed in
reboundx/test/test_machine_independent.py).  They are data, copied byte for byte."""
import glob
import os
import warnings

import pytest

from reboundx_amd.extras import Extras, REBX_INTEGRATORS, Simulation, lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rebx")
FORCE_FILES = ["central_force", "gr", "gr_full", "gr_potential", "gravitational_harmonics", "modify_orbits_forces",
               "radiation_forces", "tides_precession"]


def three_body(integrator="whfast"):
    sim = Simulation(integrator=integrator)
    sim.add(m=1.)
    sim.add(m=1e-4, x=1., vy=1.)
    sim.add(m=1e-4, x=2., vy=0.7)
    return sim


def load(name, sim=None):
    sim = sim or three_body()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        rebx = Extras(sim, os.path.join(GOLDEN, name + ".rebx"))
    return sim, rebx, [str(w.message) for w in caught]


# what reboundx/test/test_machine_independent.py stored in each file
EXPECTED = {
    "gr": ({"c": 1e4}, {}),
    "gr_full": ({"c": 1e4}, {}),
    "gr_potential": ({"c": 1e4}, {}),
    "radiation_forces": ({"c": 1e4}, {1: {"beta": 0.5}, 2: {"beta": 0.3}}),
    "tides_precession": ({}, {0: {"R_tides": 1e-3, "k1": 0.4}}),
    "central_force": ({}, {0: {"Acentral": 1e-3, "gammacentral": -1.0}}),
    "gravitational_harmonics": ({}, {0: {"J2": 1e-3, "J4": 1e-3, "R_eq": 1e-3}}),
    "modify_orbits_forces": ({}, {1: {"tau_a": -1e4, "tau_e": -1e3}, 2: {"tau_e": -1e3, "tau_inc": -1e3}}),
}


@pytest.mark.parametrize("name", FORCE_FILES)
def test_reference_fixture_loads_with_its_force_and_parameters(name):
    sim, rebx, msgs = load(name)
    assert len(msgs) == 1 and "different version" in msgs[0]        # written by 2.19.3
    force = rebx.get_force(name)
    assert sim.additional_forces                                     # re-added to the simulation
    assert lib.rebx_len(rebx._additional_forces) == 1 and lib.rebx_len(rebx._allocated_forces) == 1
    fp, pp = EXPECTED[name]
    for k, v in fp.items():
        assert force.params[k] == v
    assert len(force.params) == len(fp)
    for i in range(3):
        want = pp.get(i, {})
        assert len(sim.particles[i].params) == len(want)
        for k, v in want.items():
            assert sim.particles[i].params[k] == v
    sim.free_particle_params()


@pytest.mark.parametrize("name", FORCE_FILES + ["test", "modify_mass", "modify_orbits_direct"])
def test_reading_then_writing_a_reference_fixture_reproduces_its_bytes(name, tmp_path):
    """Everything after the 64-byte header (which names the writing version) must come back
    byte for byte: record order, sizes, list order, values.  test.rebx (two particles) holds an
    integrate_force operator whose force parameter is stored by name, and a pre-timestep step."""
    sim = three_body()
    if name == "test":
        sim = Simulation()
        sim.add(m=1.)
        sim.add(m=1e-4, x=1., vy=1.)
    sim, rebx, _ = load(name, sim)
    out = str(tmp_path / "again.rebx")
    rebx.save(out)
    a = open(os.path.join(GOLDEN, name + ".rebx"), "rb").read()
    b = open(out, "rb").read()
    assert b[:31] == b"REBOUNDx Binary File. Version: " and b[31:36] == b"3.0.4" and len(b[:64]) == 64
    assert a[64:] == b[64:]
    sim.free_particle_params()


def test_fixture_whose_step_names_a_missing_operator_loads_the_rest():
    """backup.rebx: a post-timestep step that refers to an operator called gr_full, which is not in
    the file's operator list; the reference drops the step with a warning and keeps the force."""
    sim, rebx, msgs = load("backup")
    assert any("operator step was not loaded" in m for m in msgs) and any("operator was not loaded" in m for m in msgs)
    assert rebx.get_force("gr_full").params["c"] == 1e4
    assert lib.rebx_len(rebx._post_timestep_modifications) == 0


def test_operator_fixtures_load_with_their_steps_and_parameters():
    """modify_mass.rebx / modify_orbits_direct.rebx: the operator, half a step before and half a
    step after (written by rebx_add_operator under WHFast), and the particles' timescales."""
    sim, rebx, _ = load("modify_mass")
    assert rebx.get_operator("modify_mass").operator_type == 1
    assert lib.rebx_len(rebx._pre_timestep_modifications) == 1 and lib.rebx_len(rebx._post_timestep_modifications) == 1
    assert sim.particles[0].params["tau_mass"] == -1e4
    sim2, rebx2, _ = load("modify_orbits_direct")
    assert rebx2.get_operator("modify_orbits_direct").name == b"modify_orbits_direct"
    assert sim2.particles[1].params["tau_omega"] == 100.0 and sim2.particles[2].params["tau_inc"] == -1e3
    assert sim2.pre_timestep_modifications and sim2.post_timestep_modifications


def test_missing_and_truncated_files_are_errors(tmp_path):
    with pytest.raises(RuntimeError, match="Cannot open binary file"):
        Extras(three_body(), str(tmp_path / "nope.rebx"))
    data = open(os.path.join(GOLDEN, "gr.rebx"), "rb").read()
    for cut in (40, 64, 100, len(data) - 10):
        p = tmp_path / ("cut%d.rebx" % cut)
        p.write_bytes(data[:cut])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(RuntimeError, match="unreadable"):
                Extras(three_body(), str(p))


def test_round_trip_of_forces_operators_steps_and_every_value_type(tmp_path):
    sim = three_body()
    rebx = Extras(sim)
    rebx.register_param("my_count", "REBX_TYPE_UINT32")
    gr = rebx.load_force("gr")
    gr.params["c"] = 123.5
    gr.params["max_iterations"] = 7
    rebx.add_force(gr)
    mof = rebx.load_force("modify_orbits_forces")
    mof.params["coordinates"] = 2
    op = rebx.load_operator("integrate_force")
    op.params["force"] = mof
    op.params["integrator"] = REBX_INTEGRATORS["rk2"]
    rebx.add_operator(op)                                            # whfast: dt/2 before and after
    sim.particles[1].params["tau_a"] = -1e5
    sim.particles[1].params["primary"] = 1
    sim.particles[2].params["my_count"] = 4000000000
    path = str(tmp_path / "state.rebx")
    rebx.save(path)

    sim2 = three_body()
    rebx2 = Extras(sim2, path)                                       # same version: no warning
    assert rebx2.get_force("gr").params["c"] == 123.5 and rebx2.get_force("gr").params["max_iterations"] == 7
    assert lib.rebx_len(rebx2._additional_forces) == 1
    op2 = rebx2.get_operator("integrate_force")
    assert op2.params["integrator"] == REBX_INTEGRATORS["rk2"]
    assert op2.params["force"].name == b"modify_orbits_forces"      # linked up by name
    assert op2.params["force"].params["coordinates"] == 2
    assert lib.rebx_len(rebx2._pre_timestep_modifications) == 1 and lib.rebx_len(rebx2._post_timestep_modifications) == 1
    assert sim2.pre_timestep_modifications and sim2.post_timestep_modifications
    assert sim2.particles[1].params["tau_a"] == -1e5 and sim2.particles[1].params["primary"] == 1
    assert sim2.particles[2].params["my_count"] == 4000000000
    # a second generation is byte-identical to the first
    again = str(tmp_path / "state2.rebx")
    rebx2.save(again)
    assert open(path, "rb").read() == open(again, "rb").read()
    sim.free_particle_params()
    sim2.free_particle_params()


def test_parameter_columns_are_saved_as_the_particle_parameters_they_stand_for(tmp_path):
    """A per-particle parameter given as a whole column (rebxamd_set_param_column) has no list node; the file stores its
    entries as ordinary parameters of the particles (output.c:209-256 layout), so whoever reads the file -- this library or
    the reference -- finds them in the particles' lists.  Unset entries are not written; a list entry of the same name wins."""
    import numpy as np
    sim = three_body()
    rebx = Extras(sim)
    rebx.add_force(rebx.load_force("modify_orbits_forces"))
    assert rebx.set_param_column("tau_a", np.array([np.nan, -1e4, -2e4]))
    assert rebx.set_param_column("beta", np.array([np.nan, np.nan, 0.25]))
    sim.particles[2].params["tau_a"] = -7e4                           # the list's own entry wins over the column's
    sim.particles[1].params["tau_e"] = -1e3
    path = str(tmp_path / "columns.rebx")
    rebx.save(path)

    sim2 = three_body()
    rebx2 = Extras(sim2, path)
    assert rebx2.get_param_column("tau_a") is None                   # they come back as list entries
    ps = sim2.particles
    assert len(ps[0].params) == 0
    assert ps[1].params["tau_a"] == -1e4 and ps[1].params["tau_e"] == -1e3 and len(ps[1].params) == 2
    assert ps[2].params["tau_a"] == -7e4 and ps[2].params["beta"] == 0.25 and len(ps[2].params) == 2
    sim.free_particle_params()
    sim2.free_particle_params()
