"""Prints error statistics of the HIP path against the CPU checker for a set of cases
(development aid; the assertions live in tests/test_gpu_parity.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from gpu_common import rel_err, run_device, run_oracle, ulp_diff  # noqa: E402
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd.device import Context  # noqa: E402

C = synthetic.C_DEFAULT
ctx = Context(0)


def report(name, cols, pp, forces, G=1.0, **kw):
    ref = run_oracle(cols, pp, forces, G)
    for mode, label in ((1, "exact"), (0, "fast")):
        try:
            got, dp = run_device(ctx, cols, pp, forces, G, mode, **kw)
        except Exception as e:  # noqa: BLE001
            print("%-28s %-5s ERROR %s" % (name, label, e))
            continue
        u = ulp_diff(got, ref).max(axis=0)
        r = rel_err(got, ref)
        nbad = int((u > 0).sum())
        print("%-28s %-5s n=%-8d rows!=: %-7d max ulp %-10d (at %d) max rel %.3e (at %d) head ulp %s" % (
            name, label, got.shape[1], nbad, u.max(), int(u.argmax()), r.max(), int(r.argmax()), u[:min(4, len(u))]))
    w = ctx.fetch_warnings()
    if w:
        print("   warnings:", w)


def zero_acc(cols):
    c = dict(cols)
    for k in ("ax", "ay", "az"):
        c[k] = np.zeros_like(cols["x"])
    return c


for n, npl in ((3000, 0), (3001, 2), (4098, 5)):
    cols, pp, _ = synthetic.disk(n, npl)
    stack = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr", {"c": C})]
    report("C2 stack zero-acc n=%d" % n, zero_acc(cols), pp, stack)
    synthetic.add_newtonian(cols, n_head=1 + npl)
    report("C2 stack newt-acc", cols, pp, stack)
    for f in stack:
        report("  " + f[0], zero_acc(cols), pp, [f])
    for coords in (0, 1, 2):
        p2 = dict(pp)
        p2["primary"] = np.zeros(n, dtype=np.uint8); p2["primary"][0] = 1
        report("  mof coords=%d" % coords, zero_acc(cols), p2, [("modify_orbits_forces", {"coordinates": coords})])
    report("  gr_potential", zero_acc(cols), pp, [("gr_potential", {"c": C})])
    report("  tides", zero_acc(cols), pp, [("tides_precession", {})])
    p3 = dict(pp); p3["Acentral"] = np.full(n, np.nan); p3["gammacentral"] = np.full(n, np.nan)
    p3["Acentral"][0] = 1e-4; p3["gammacentral"][0] = -1.0
    report("  central", zero_acc(cols), p3, [("central_force", {})])

# general regime: every particle massive
n = 20001
cols, pp, _ = synthetic.disk(n, 1, dust_mass=1e-15)
p2 = dict(pp); p2["primary"] = np.zeros(n, dtype=np.uint8); p2["primary"][0] = 1
stack = [("tides_precession", {}), ("modify_orbits_forces", {"coordinates": 2}), ("gr_potential", {"c": C})]
report("C3 stack general", zero_acc(cols), p2, stack)
for f in stack:
    report("  " + f[0], zero_acc(cols), p2, [f])
report("  harmonics general", zero_acc(cols), p2, [("gravitational_harmonics", {})])

# all-massive small systems
tp = synthetic.twoplanets()
n = 3
nan = np.full(n, np.nan)
pp = {"beta": np.array([np.nan, .5, .3]), "J2": np.array([1e-3, np.nan, np.nan]), "J4": np.array([1e-3, np.nan, np.nan]),
      "R_eq": np.array([1e-3, np.nan, np.nan]), "R_tides": np.array([1e-3, np.nan, np.nan]), "k1": np.array([.4, np.nan, np.nan]),
      "tau_a": np.array([np.nan, -1e4, np.nan]), "tau_e": np.array([np.nan, -1e3, -1e3]), "tau_inc": np.array([np.nan, np.nan, -1e3]),
      "Acentral": np.array([1e-3, np.nan, np.nan]), "gammacentral": np.array([-1., np.nan, np.nan])}
for f in (("radiation_forces", {"c": 1e4}), ("gravitational_harmonics", {}), ("gr", {"c": 1e4}), ("gr_potential", {"c": 1e4}),
          ("tides_precession", {}), ("central_force", {}), ("modify_orbits_forces", {}), ("modify_orbits_forces", {"coordinates": 1})):
    report("twoplanets " + f[0], tp, pp, [f])

# gr_full
for n in (2, 3, 16, 128, 300):
    cl = synthetic.cluster(n)
    s = oracle.Sim(cl); oracle.newtonian_gravity(s)
    cl["ax"], cl["ay"], cl["az"] = [a.copy() for a in s.acc]
    for ig in (0, 1):
        ref = oracle.Sim(cl); sw = oracle.gr_full(ref, 1e2, ig)
        for mode, label in ((1, "exact"), (0, "fast")):
            from reboundx_amd.device import DeviceParticles
            dp = DeviceParticles.from_host(cl, G=1.0)
            sw2 = ctx.grfull(dp.view(), 1e2, ig, mode)
            got = np.array(dp.acc_host())
            d_ref = np.array(ref.acc) - np.array([cl["ax"], cl["ay"], cl["az"]])
            d_got = got - np.array([cl["ax"], cl["ay"], cl["az"]])
            print("gr_full n=%-4d ig=%d %-5s sweeps ref %d dev %d  max ulp(total) %d  max rel(delta) %.3e" % (
                n, ig, label, sw, sw2, ulp_diff(got, np.array(ref.acc)).max(), rel_err(d_got, d_ref).max()))
print("done")
