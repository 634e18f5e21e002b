"""Synthetic workloads: debris disks and clusters, generated from a counter-based RNG so
that any index range can be produced independently (each rank of a sharded run builds only
its own particles, and CPU and GPU sides see identical inputs).

Units: G = 1; speed of light 10065.32 (AU, yr/2pi, Msun), the value the reference's
``reboundx/constants.py:5`` lists.
"""
import numpy as np

C_DEFAULT = 10065.32012038219
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed, index, field):
    """U[0,1) keyed by (seed, particle index, field)."""
    with np.errstate(over="ignore"):
        idx = np.asarray(index, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(field))
        z = _splitmix64(idx ^ key)
        z = _splitmix64(z + key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def kepler_to_cartesian(mu, a, e, inc, Omega, omega, f):
    """Orbital elements -> position and velocity relative to the primary."""
    r = a * (1.0 - e * e) / (1.0 + e * np.cos(f))
    v0 = np.sqrt(mu / (a * (1.0 - e * e)))
    cO, sO, co, so, ci, si, cf, sf = np.cos(Omega), np.sin(Omega), np.cos(omega), np.sin(omega), np.cos(inc), np.sin(inc), np.cos(f), np.sin(f)
    x = r * (cO * (co * cf - so * sf) - sO * (so * cf + co * sf) * ci)
    y = r * (sO * (co * cf - so * sf) + cO * (so * cf + co * sf) * ci)
    z = r * (so * cf + co * sf) * si
    vx = v0 * ((e + cf) * (-ci * co * sO - cO * so) - sf * (co * cO - ci * so * sO))
    vy = v0 * ((e + cf) * (ci * co * cO - sO * so) - sf * (co * sO + ci * so * cO))
    vz = v0 * ((e + cf) * co * si - sf * si * so)
    return x, y, z, vx, vy, vz


def disk(n, n_planets=0, seed=3, lo=0, hi=None, dust_mass=0.0, with_head=True):
    """Star (m=1, index 0) + ``n_planets`` planets (m=1e-3, a=5.2k, e=0.05) + dust on
    a in [1,1.2], e in [0,0.1], inc in [0,0.1].  Returns the particles with global index in
    [lo,hi) -- preceded by the head (star + planets) when ``with_head`` and lo > head -- as a
    dict of numpy columns, plus per-particle parameter columns:
      beta in [0.01,0.5] on dust; star J2=J4=R_eq=1e-3, R_tides=1e-3, k1=0.4;
      dust tau_a=-1e4, tau_e=-1e3, tau_inc=-1e3, R_tides=1e-5, k1=0.3.
    The star sits at rest at the origin (an inertial frame in which the head's centre of
    mass moves slightly is as good as any for force evaluation)."""
    hi = n if hi is None else hi
    head = 1 + n_planets
    idx_parts = []
    if with_head and lo > 0:
        idx_parts.append(np.arange(0, min(head, n), dtype=np.int64))
        lo = max(lo, head)
    idx_parts.append(np.arange(lo, hi, dtype=np.int64))
    idx = np.concatenate(idx_parts)
    k = len(idx)
    cols = {c: np.zeros(k) for c in ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")}
    nan = np.full(k, np.nan)
    pp = {c: nan.copy() for c in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc", "J2", "J4", "R_eq")}

    star = idx == 0
    planet = (idx >= 1) & (idx < head)
    dust = idx >= head
    cols["m"][star] = 1.0
    cols["m"][planet] = 1.0e-3
    cols["m"][dust] = dust_mass

    def elems(sel, a, e, inc):
        i = idx[sel]
        Om = 2 * np.pi * uniform(seed, i, 3)
        om = 2 * np.pi * uniform(seed, i, 4)
        f = 2 * np.pi * uniform(seed, i, 5)
        return kepler_to_cartesian(1.0 + cols["m"][sel], a, e, inc, Om, om, f)

    if planet.any():
        i = idx[planet]
        vals = elems(planet, 5.2 * i, np.full(len(i), 0.05), 0.01 * uniform(seed, i, 2))
        for c, v in zip(("x", "y", "z", "vx", "vy", "vz"), vals):
            cols[c][planet] = v
    if dust.any():
        i = idx[dust]
        a = 1.0 + 0.2 * uniform(seed, i, 0)
        e = 0.1 * uniform(seed, i, 1)
        inc = 0.1 * uniform(seed, i, 2)
        vals = elems(dust, a, e, inc)
        for c, v in zip(("x", "y", "z", "vx", "vy", "vz"), vals):
            cols[c][dust] = v
        pp["beta"][dust] = 0.01 + 0.49 * uniform(seed, i, 6)
        pp["tau_a"][dust] = -1.0e4
        pp["tau_e"][dust] = -1.0e3
        pp["tau_inc"][dust] = -1.0e3
        pp["R_tides"][dust] = 1.0e-5
        pp["k1"][dust] = 0.3
    pp["J2"][star] = 1.0e-3
    pp["J4"][star] = 1.0e-3
    pp["R_eq"][star] = 1.0e-3
    pp["R_tides"][star] = 1.0e-3
    pp["k1"][star] = 0.4
    return cols, pp, idx


def disk_device(n, device, dust_mass=0.0, seed=3, newtonian=False):
    """The same kind of disk as :func:`disk` with no planets (star at index 0, dust on a in [1,1.2], e in
    [0,0.1], inc in [0,0.1], the same parameter values), generated ON THE GPU with torch's generator, for
    sizes at which building the columns on the host would dominate a run (1e8 bodies: 50 ms instead of 30 s).
    Not the same random numbers as :func:`disk`: callers that need a host copy of some bodies copy those rows
    back.  Returns a dict of float64 device columns: x y z vx vy vz m ax ay az beta R_tides k1 tau_a tau_e
    tau_inc (NaN = not set).  newtonian: accelerations start as the star's Newtonian pull instead of zero."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    u = lambda: torch.rand(n, dtype=torch.float64, device=device, generator=g)
    a = 1.0 + 0.2 * u()
    e = 0.1 * u()
    inc = 0.1 * u()
    Om, om, f = 2 * np.pi * u(), 2 * np.pi * u(), 2 * np.pi * u()
    mu = 1.0 + dust_mass
    r = a * (1.0 - e * e) / (1.0 + e * torch.cos(f))
    v0 = torch.sqrt(mu / (a * (1.0 - e * e)))
    del a
    cO, sO, co, so, ci, si, cf, sf = torch.cos(Om), torch.sin(Om), torch.cos(om), torch.sin(om), torch.cos(inc), torch.sin(inc), torch.cos(f), torch.sin(f)
    del Om, om, inc, f
    cols = {}
    cols["x"] = r * (cO * (co * cf - so * sf) - sO * (so * cf + co * sf) * ci)
    cols["y"] = r * (sO * (co * cf - so * sf) + cO * (so * cf + co * sf) * ci)
    cols["z"] = r * (so * cf + co * sf) * si
    cols["vx"] = v0 * ((e + cf) * (-ci * co * sO - cO * so) - sf * (co * cO - ci * so * sO))
    cols["vy"] = v0 * ((e + cf) * (ci * co * cO - sO * so) - sf * (co * sO + ci * so * cO))
    cols["vz"] = v0 * ((e + cf) * co * si - sf * si * so)
    del r, v0, e, cO, sO, co, so, ci, si, cf, sf
    cols["m"] = torch.full((n,), float(dust_mass), dtype=torch.float64, device=device)
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        cols[k][0] = 0.0
    cols["m"][0] = 1.0
    for k in ("ax", "ay", "az"):
        cols[k] = torch.zeros(n, dtype=torch.float64, device=device)
    if newtonian:
        r2 = cols["x"] ** 2 + cols["y"] ** 2 + cols["z"] ** 2
        r2[0] = 1.0
        w = -1.0 / (r2 * torch.sqrt(r2))
        w[0] = 0.0
        for k, s in (("ax", "x"), ("ay", "y"), ("az", "z")):
            cols[k] = w * cols[s]
        del r2, w
    cols["beta"] = 0.01 + 0.49 * u()
    for k, val, star in (("tau_a", -1.0e4, np.nan), ("tau_e", -1.0e3, np.nan), ("tau_inc", -1.0e3, np.nan), ("R_tides", 1.0e-5, 1.0e-3), ("k1", 0.3, 0.4)):
        cols[k] = torch.full((n,), val, dtype=torch.float64, device=device)
        cols[k][0] = star
    cols["beta"][0] = np.nan
    return cols


def star_params():
    """Per-particle parameters of the star of :func:`disk` / :func:`disk_device`, as length-1 host columns
    (what :func:`reboundx_amd.program.build_ops` needs when every source is the star)."""
    nan = np.array([np.nan])
    return {"beta": nan.copy(), "tau_a": nan.copy(), "tau_e": nan.copy(), "tau_inc": nan.copy(), "R_tides": np.array([1.0e-3]), "k1": np.array([0.4]),
            "J2": np.array([1.0e-3]), "J4": np.array([1.0e-3]), "R_eq": np.array([1.0e-3]), "primary": np.array([1], dtype=np.uint8)}


def add_newtonian(cols, G=1.0, n_head=None):
    """Adds the Newtonian pull of the massive head bodies to the accelerations (what REBOUND's
    gravity routine leaves in particles[i].a before the additional-force hook runs)."""
    m = cols["m"]
    n_head = len(m) if n_head is None else n_head
    for u in range(min(n_head, len(m))):
        if m[u] == 0.0:
            continue
        dx, dy, dz = cols["x"][u] - cols["x"], cols["y"][u] - cols["y"], cols["z"][u] - cols["z"]
        r2 = dx * dx + dy * dy + dz * dz
        r2[u] = 1.0
        w = G * m[u] / (r2 * np.sqrt(r2))
        w[u] = 0.0
        cols["ax"] += w * dx
        cols["ay"] += w * dy
        cols["az"] += w * dz
    return cols


def cluster(n, seed=3, c=1.0e4):
    """Uniform-density sphere of radius 1, equal masses 1/n, isotropic velocities at roughly
    the virial dispersion; G=1.  Accelerations are left zero (callers add Newtonian gravity)."""
    i = np.arange(n, dtype=np.int64)
    r = np.cbrt(uniform(seed, i, 0))
    ct = 2 * uniform(seed, i, 1) - 1
    ph = 2 * np.pi * uniform(seed, i, 2)
    st = np.sqrt(1 - ct * ct)
    cols = {"x": r * st * np.cos(ph), "y": r * st * np.sin(ph), "z": r * ct, "m": np.full(n, 1.0 / n)}
    sigma = np.sqrt(0.2)      # sqrt(GM/(5R)) per component for a uniform sphere in virial equilibrium
    for k, (f1, f2) in enumerate(((3, 4), (5, 6), (7, 8))):
        u1 = np.maximum(uniform(seed, i, f1), 1e-300)
        u2 = uniform(seed, i, f2)
        cols[("vx", "vy", "vz")[k]] = sigma * np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2)
    for k in ("ax", "ay", "az"):
        cols[k] = np.zeros(n)
    return cols


def twoplanets():
    """The system the reference's tests use (test_machine_independent.py:25-31): star m=1,
    planets m=1e-4 at a=1 (e=0.1) and a=2 (e=0.1, inc=0.2), moved to the centre of mass."""
    m = np.array([1.0, 1.0e-4, 1.0e-4])
    x = np.zeros(3); y = np.zeros(3); z = np.zeros(3); vx = np.zeros(3); vy = np.zeros(3); vz = np.zeros(3)
    # REBOUND adds particles in Jacobi coordinates: each orbit is about the centre of mass of the bodies added so far
    for k, (a, e, inc) in enumerate(((1.0, 0.1, 0.0), (2.0, 0.1, 0.2)), start=1):
        mint = m[:k].sum()
        cx = [(m[:k] * q[:k]).sum() / mint for q in (x, y, z, vx, vy, vz)]
        px, py, pz, pvx, pvy, pvz = kepler_to_cartesian(mint + m[k], a, e, inc, 0.0, 0.0, 0.0)
        x[k], y[k], z[k], vx[k], vy[k], vz[k] = cx[0] + px, cx[1] + py, cx[2] + pz, cx[3] + pvx, cx[4] + pvy, cx[5] + pvz
    M = m.sum()
    for q in (x, y, z, vx, vy, vz):
        q -= (m * q).sum() / M
    return {"x": x, "y": y, "z": z, "vx": vx, "vy": vy, "vz": vz, "m": m, "ax": np.zeros(3), "ay": np.zeros(3), "az": np.zeros(3)}
