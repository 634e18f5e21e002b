"""Operators modify_mass and modify_orbits_direct on the GPU against the CPU checker's
restatement, which reproduces the reference's operator archives (tests/test_oracle.py).

The orbit <-> Cartesian conversions go through acos/cos/sin of the device math library, which
differs from the host's by an ulp or two, and the individual angles are ill-conditioned for
nearly circular or nearly planar orbits (their sum is not), so the bar is 1e-11 of the size of
the state rather than bits."""
import numpy as np
import pytest
import torch

import oracle
from reboundx_amd import synthetic
from reboundx_amd.device import DeviceParticles

pytestmark = pytest.mark.gpu
STATE = ("x", "y", "z", "vx", "vy", "vz", "m")


@pytest.fixture(scope="module")
def ctx():
    from reboundx_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def dev(cols, pp, **kw):
    host = dict(cols)
    for k in ("tau_a", "tau_e", "tau_inc"):
        host[k] = pp[k]
    return DeviceParticles.from_host(host, device="cuda:0", **kw)


def state(dp):
    return np.array([dp.cols[k].cpu().numpy() for k in STATE])


@pytest.mark.parametrize("n,n_planets,dust_mass", [(3, 2, 0.0), (5000, 3, 0.0), (40, 39, 0.0), (100_000, 1, 0.0)])
def test_modify_mass(ctx, n, n_planets, dust_mass):
    cols, pp, _ = synthetic.disk(n, n_planets, dust_mass=dust_mass)
    tau = np.full(n, np.nan)
    tau[0] = -1e3
    if n > 2:
        tau[2] = 5e2
    ref = oracle.Sim(cols)
    oracle.modify_mass(ref, 0.7, tau)
    dp = dev(cols, pp)
    ctx.modify_mass(dp.view(), torch.tensor(tau, device="cuda:0"), 0.7)
    got, want = state(dp), np.array([ref.cols[k] for k in STATE])
    assert np.array_equal(got[6], want[6])                       # masses: elementwise, same expression
    assert np.abs(got[:6] - want[:6]).max() <= 1e-14 * np.abs(want[:6]).max()
    assert abs((got[6] * got[0]).sum()) <= 1e-14                 # now in the centre-of-mass frame


@pytest.mark.parametrize("coordinates", [0, 1, 2])
@pytest.mark.parametrize("n,n_planets", [(3, 2), (6000, 3), (50, 49)])
def test_modify_orbits_direct(ctx, n, n_planets, coordinates):
    cols, pp, _ = synthetic.disk(n, n_planets)
    rng = np.random.default_rng(n)
    tom, tOm = np.full(n, np.nan), np.full(n, np.nan)
    tom[1::3] = 100.0
    tOm[2::5] = -300.0
    tau_a, tau_e, tau_inc = pp["tau_a"].copy(), pp["tau_e"].copy(), pp["tau_inc"].copy()
    tau_a[1], tau_e[1] = -1e4, -1e3                              # planets too
    if n > 2:
        tau_e[2], tau_inc[2] = -1e3, -1e3
    pp = dict(pp, tau_a=tau_a, tau_e=tau_e, tau_inc=tau_inc)
    prim = np.zeros(n, dtype=np.uint8)
    prim[0] = 1
    ref = oracle.Sim(cols)
    assert oracle.modify_orbits_direct(ref, 0.3, coordinates, 0.7, tau_a, tau_e, tau_inc, tom, tOm, prim) == 0
    dp = dev(cols, pp, n_head=n_planets + 1, tail_massless=True)
    ctx.modify_orbits_direct(dp.view(), 0.3, coordinates, 0, 0.7, torch.tensor(tom, device="cuda:0"), torch.tensor(tOm, device="cuda:0"))
    got, want = state(dp), np.array([ref.cols[k] for k in STATE])
    assert np.isfinite(got).all()
    assert np.abs(want[:6] - np.array([cols[k] for k in STATE[:6]])).max() > 1e-5     # something happened
    assert np.abs(got[:6] - want[:6]).max() <= 1e-11 * np.abs(want[:6]).max()


def test_modify_orbits_direct_refuses_a_long_massive_chain(ctx):
    from reboundx_amd.device import RebxAmdError
    cols, pp, _ = synthetic.disk(300, 1, dust_mass=1e-9)
    dp = dev(cols, pp)
    with pytest.raises(RebxAmdError) as ei:
        ctx.modify_orbits_direct(dp.view(), 0.1)
    assert ei.value.code == 3
