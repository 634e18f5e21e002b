"""The outputs the reference's own notebooks record (tests/golden/notebook_kats.json), reproduced on the device-resident path:
the force in the kick of a Wisdom-Holman run on the GPU (rebxamd_wh_steps), two bodies, dt = P/1000 (the splitting error of
the scheme, of relative order (n dt)^2 / 12 of the effect, is then below the tolerance the CPU checker meets with an
8th-order Runge-Kutta integration of the same systems, tests/test_oracle.py).  The notebooks integrate with IAS15."""
import json
import os

import numpy as np
import pytest

from reboundx_amd import synthetic
from reboundx_amd.device import DeviceParticles
from reboundx_amd.program import build_ops

pytestmark = pytest.mark.gpu
STATE = ("x", "y", "z", "vx", "vy", "vz")
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_kats.json")))


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def two_body(m1, a, e):
    m = np.array([1.0, m1])
    x, y, z, vx, vy, vz = synthetic.kepler_to_cartesian(1.0 + m1, a, e, 0.0, 0.0, 0.0, 0.0)
    cols = {"x": np.array([0.0, x]), "y": np.array([0.0, y]), "z": np.array([0.0, z]),
            "vx": np.array([0.0, vx]), "vy": np.array([0.0, vy]), "vz": np.array([0.0, vz]), "m": m}
    for k in STATE:
        cols[k] = cols[k] - (m * cols[k]).sum() / m.sum()
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(2)
    return cols, 2 * np.pi * np.sqrt(a ** 3 / m.sum())


def pomega(dp, m):
    s = np.array([dp.cols[k].cpu().numpy() for k in STATE])
    r, v = s[:3, 1] - s[:3, 0], s[3:, 1] - s[3:, 0]
    mu = m.sum()
    e = ((v @ v - mu / np.sqrt(r @ r)) * r - (r @ v) * v) / mu
    return float(np.arctan2(e[1], e[0]))


def run(ctx, dp, ops, span, period, mode):
    n = int(np.ceil(span / (period / 1000.0)))
    ctx.wh_steps(dp.view(), ops, span / n, n, mode)


@pytest.mark.parametrize("mode", [1, 0], ids=["exact", "fast"])
def test_gr_notebook_pericentre_advance(ctx, mode):
    """GeneralRelativity.ipynb cells 1-9: ten time units of gravity alone, then a hundred with gr."""
    cols, period = two_body(1.66013e-07, 0.387098, 0.205630)
    dp = DeviceParticles.from_host(cols, device="cuda:0", n_head=2, tail_massless=True)
    run(ctx, dp, [], 10.0, period, mode)
    assert abs(pomega(dp, cols["m"])) < 1e-10
    run(ctx, dp, build_ops([("gr", {"c": synthetic.C_DEFAULT})], {}), 100.0, period, mode)
    assert pomega(dp, cols["m"]) == pytest.approx(float(KATS["gr_pomega"]["pomega_t110"]), abs=2e-9)


def test_gr_from_the_start_notebook_values(ctx):
    """SavingAndLoadingSimulations.ipynb cells 1-12: the same system with gr from t = 0, read at t = 10 and t = 20."""
    cols, period = two_body(1.66013e-07, 0.387098, 0.205630)
    dp = DeviceParticles.from_host(cols, device="cuda:0", n_head=2, tail_massless=True)
    ops = build_ops([("gr", {"c": synthetic.C_DEFAULT})], {})
    run(ctx, dp, ops, 10.0, period, 0)
    assert pomega(dp, cols["m"]) == pytest.approx(float(KATS["gr_pomega_from_t0"]["pomega_t10"]), abs=1e-9)
    run(ctx, dp, ops, 10.0, period, 0)
    assert pomega(dp, cols["m"]) == pytest.approx(float(KATS["gr_pomega_from_t0"]["pomega_t20"]), abs=1e-9)


def test_harmonics_notebook_pericentre_advance(ctx):
    """J2.ipynb cells 1-6: star with J2 = 0.1, J4 = 0.2, R_eq = 1/200; gravity alone to t = 1, harmonics to t = 1e4."""
    cols, period = two_body(1.0e-5, 1.0, 1.0e-2)
    dp = DeviceParticles.from_host(cols, device="cuda:0", n_head=2, tail_massless=True)
    run(ctx, dp, [], 1.0, period, 0)
    nan = np.array([np.nan, np.nan])
    J2, J4, Req = nan.copy(), nan.copy(), nan.copy()
    J2[0], J4[0], Req[0] = 0.1, 0.2, 1 / 200.0
    ops = build_ops([("gravitational_harmonics", {})], {"J2": J2, "J4": J4, "R_eq": Req})
    run(ctx, dp, ops, 1.0e4 - 1.0, period, 0)
    assert pomega(dp, cols["m"]) == pytest.approx(float(KATS["j2_pomega"]["pomega_t1e4"]), abs=2e-7)
