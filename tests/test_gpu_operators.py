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


def chain_case(layout, n, seed=7):
    """Layouts the short-prefix path cannot take: every body massive, or planets scattered behind massless dust."""
    if layout == "all-massive":
        cols, pp, _ = synthetic.disk(n, 2, dust_mass=3e-8, seed=seed)
    elif layout == "heavy-dust":                    # total dust mass 0.3 of the star's: strong coupling between the steps
        cols, pp, _ = synthetic.disk(n, 2, dust_mass=0.3 / n, seed=seed)
    else:                                           # star first, planets anywhere among massless dust
        cols, pp, _ = synthetic.disk(n, 5, seed=seed)
        perm = np.concatenate([[0], 1 + np.random.default_rng(seed).permutation(n - 1)])
        cols = {k: np.ascontiguousarray(v[perm]) for k, v in cols.items()}
        pp = {k: np.ascontiguousarray(v[perm]) for k, v in pp.items()}
    massive = np.nonzero(cols["m"])[0]
    tau_a, tau_e, tau_inc = pp["tau_a"].copy(), pp["tau_e"].copy(), pp["tau_inc"].copy()
    for j, i in enumerate(massive[1:6]):            # the heavy bodies migrate too
        tau_a[i], tau_e[i], tau_inc[i] = -1e4 / (1 + j), -1e3, -2e3
    tom, tOm = np.full(n, np.nan), np.full(n, np.nan)
    tom[1::3] = 100.0
    tOm[2::5] = -300.0
    return cols, dict(pp, tau_a=tau_a, tau_e=tau_e, tau_inc=tau_inc), tom, tOm


@pytest.mark.parametrize("coordinates", [0, 1, 2])
@pytest.mark.parametrize("layout,n", [("all-massive", 70), ("all-massive", 3000), ("heavy-dust", 2500), ("planets-scattered", 4001)])
def test_modify_orbits_direct_with_massive_bodies_anywhere(ctx, layout, n, coordinates):
    """The reference's chain (rebxtools.c:164-214: every massive body, last to first, shifts the bodies before it
    ahead of their own steps) beyond a 64-body prefix: found on the device as the fixed point of a parallel map,
    checked against the checker's sequential walk.

    Bar: 1e-12 of the size of the state, plus what the reference's own answer moves by when its inputs move by one
    ulp (measured here with the checker, three random perturbations).  That second term is negligible in Jacobi and
    PARTICLE coordinates (1e-14 .. 1e-11) and dominates in barycentric ones (1e-10 .. 3e-9): there the star is
    stepped too, about a barycentre 1e-4 away from it, the angles of that orbit come out of acos near +-1, and the
    round trip's error is handed to every body with a mass ratio of nearly one.  The sequential fallback on the
    device must agree to the same bar."""
    cols, pp, tom, tOm = chain_case(layout, n)
    primary = 0
    if coordinates == 2 and layout == "planets-scattered":      # a primary that is not the first particle, beyond any short prefix
        primary = 171
        for d in (cols, pp):
            for k in d:
                d[k][[0, primary]] = d[k][[primary, 0]]
        tom[[0, primary]] = tom[[primary, 0]]
        tOm[[0, primary]] = tOm[[primary, 0]]
    prim = np.zeros(n, dtype=np.uint8)
    prim[primary] = 1
    dt = 0.3

    def checker(c):
        ref = oracle.Sim(c)
        assert oracle.modify_orbits_direct(ref, dt, coordinates, 0.7, pp["tau_a"], pp["tau_e"], pp["tau_inc"], tom, tOm, prim) == 0
        return np.array([ref.cols[k] for k in STATE])

    want = checker(cols)
    start = np.array([cols[k] for k in STATE])
    assert np.isfinite(want).all() and np.abs(want[:6] - start[:6]).max() > 1e-5
    rng = np.random.default_rng(1)
    floor = 0.
    for _ in range(3):
        moved = {k: v.copy() for k, v in cols.items()}
        for k in STATE[:6]:
            moved[k] = np.nextafter(moved[k], moved[k] + rng.choice([-1., 1.], n))
        floor = max(floor, np.abs(checker(moved)[:6] - want[:6]).max())
    scale = np.abs(want[:6]).max()
    bar = 1e-12 * scale + 4. * floor
    assert bar <= (1e-7 if coordinates == 1 else 1e-10)
    results = {}
    for label, rounds in (("rounds", 12), ("walk", 0)):
        if label == "walk" and coordinates == 2:
            continue                                # PARTICLE coordinates have no chain: one parallel pass
        dp = dev(cols, pp)
        ctx.direct_chain(rounds)
        ctx.modify_orbits_direct(dp.view(), dt, coordinates, primary, 0.7, torch.tensor(tom, device="cuda:0"), torch.tensor(tOm, device="cuda:0"))
        took = ctx.direct_chain(12)
        if label == "walk":
            assert took == -1
        elif coordinates == 2:
            assert took == 1
        else:
            assert 2 <= took <= 11, took
        got = state(dp)
        assert np.isfinite(got).all()
        assert np.array_equal(got[6], start[6])
        err = np.abs(got[:6] - want[:6]).max()
        assert err <= bar, (label, err, bar, floor)
        results[label] = got
    if "walk" in results:
        assert np.abs(results["rounds"][:6] - results["walk"][:6]).max() <= bar
    # what the bodies' steps hand out the others take back: Jacobi and PARTICLE steps leave the centre of mass where it was
    if coordinates != 1:
        m = start[6]
        for k in range(6):
            assert abs((m * results["rounds"][k]).sum() - (m * start[k]).sum()) <= 1e-13 * (m * np.abs(start[k])).sum() + 1e-18


def test_modify_orbits_direct_long_chain_at_300000_massive_grains(ctx):
    """Size the checker's O(N^2) walk cannot reach: 3e5 grains of 1e-12, Jacobi coordinates.  The rounds must settle,
    agree with the sequential walk on the device (same expressions, running sums instead of scans), and keep the
    centre of mass."""
    n = 300_000
    cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-12)
    dp = dev(cols, pp)
    ctx.direct_chain(12)
    ctx.modify_orbits_direct(dp.view(), 0.3)
    took = ctx.direct_chain(-1)
    assert 1 <= took <= 4, took
    got = state(dp)
    dq = dev(cols, pp)
    ctx.direct_chain(0)
    ctx.modify_orbits_direct(dq.view(), 0.3)
    assert ctx.direct_chain(12) == -1
    walk = state(dq)
    start = np.array([cols[k] for k in STATE])
    assert np.abs(got[:6] - start[:6]).max() > 1e-6
    assert np.abs(got[:6] - walk[:6]).max() <= 1e-13 * np.abs(walk[:6]).max()
    m = start[6]
    for k in range(6):
        assert abs((m * got[k]).sum() - (m * start[k]).sum()) <= 1e-13 * (m * np.abs(start[k])).sum()
