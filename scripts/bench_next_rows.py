"""Times the rows that sit on either side of the force pass (development aid): the
integrate_force schemes, kick/drift, the potential reductions, the gr Hamiltonian and the
prefix-sum path of gr / Jacobi modify_orbits_forces.
usage: python scripts/bench_next_rows.py [n] [reps]

Algorithmic bytes per particle (fp64 columns, C5 stack).  With massless bodies behind the star
one pass evaluates every stage of a scheme in registers: positions, velocities, mass and the six
parameters read (104 B), velocities and first-stage accelerations written (48 B) = 152 B for
euler, rk2 and rk4 alike.  As separate passes (CARRIED=0 in the environment; 152 B per force
pass with the accelerations read and written):
  euler  = 1 pass + kick (v r/w 48 + a 24)                                          = 224 B
  rk2    = 2 passes + stage (v 24 + a 24 + out 24) + final (v 48 + 2a 48)           = 472 B
  rk4    = 4 passes + 2 stages + mid (v 24, k3 48, k2 48, out 24) + final (v 48 + 3a 72) = 1016 B
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reboundx_amd import synthetic  # noqa: E402
from reboundx_amd._lib import OP_GR_POTENTIAL, OP_J2, OP_J4, OP_TIDES, Op  # noqa: E402
from reboundx_amd.device import Context, DeviceParticles  # noqa: E402
from reboundx_amd.program import build_ops  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C = synthetic.C_DEFAULT
ctx = Context(0)
CARRIED = os.environ.get("CARRIED") != "0"
ctx.carried_steps(CARRIED)
C5 = [("radiation_forces", {"c": C}), ("gravitational_harmonics", {}), ("gr_potential", {"c": C}), ("tides_precession", {}),
      ("modify_orbits_forces", {"coordinates": 2})]


def timed(fn, reps=reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


cols, pp, _ = synthetic.disk(n, 0)
pp["primary"] = np.zeros(n, dtype=np.uint8); pp["primary"][0] = 1
host = dict(cols)
for k in ("beta", "R_tides", "k1", "tau_a", "tau_e", "tau_inc"):
    host[k] = pp[k]
dp = DeviceParticles.from_host(host)
ops = build_ops(C5, {k: v[:1] for k, v in pp.items()})
view = dp.view()
print("n = %d, C5 stack (5 effects)" % n)
t = timed(lambda: ctx.apply(view, ops, 0))
print("  force pass                 %9.1f us   %6.2f TB/s algorithmic (152 B)" % (t * 1e6, 152 * n / t / 1e12))
BYTES = ({"euler": 152, "rk2": 152, "rk4": 152, "implicit_midpoint": None} if CARRIED else
         {"euler": 152 + 72, "rk2": 2 * 152 + 72 + 96, "rk4": 4 * 152 + 2 * 72 + 144 + 120, "implicit_midpoint": None})
for integ in ("euler", "rk2", "rk4", "implicit_midpoint"):
    for mode in (0, 1):
        its = []
        t = timed(lambda: its.append(ctx.integrate_force(view, ops, integ, 1e-3, mode)))
        b = BYTES[integ]
        extra = "" if b is None else "   %6.2f TB/s algorithmic (%d B)" % (b * n / t / 1e12, b)
        print("  integrate_force %-17s %-5s %9.1f us%s   %.3e particle-steps/s%s" % (
            integ, "fast" if mode == 0 else "exact", t * 1e6, extra, n / t, "" if integ != "implicit_midpoint" else "   (%d iterations)" % its[-1]))
t = timed(lambda: ctx.kick(view, 1e-3))
print("  kick                       %9.1f us   %6.2f TB/s (72 B)" % (t * 1e6, 72 * n / t / 1e12))
t = timed(lambda: ctx.drift(view, 1e-3))
print("  drift                      %9.1f us   %6.2f TB/s (72 B)" % (t * 1e6, 72 * n / t / 1e12))
del dp

# potentials and the gr Hamiltonian: every particle massive
cols, pp, _ = synthetic.disk(n, 0, dust_mass=1e-15)
host = dict(cols)
host["R_tides"], host["k1"] = pp["R_tides"], pp["k1"]
dp = DeviceParticles.from_host(host)
view = dp.view()
pops = [Op(OP_J2, 0, 0, 0, 1e-3, 1e-3), Op(OP_J4, 0, 0, 0, 1e-3, 1e-3), Op(OP_TIDES, 0, 0, 0, 0., 0.), Op(OP_GR_POTENTIAL, 0, 0, 0, C, 0.)]
t = timed(lambda: ctx.potentials(view, pops))
print("  potentials (J2,J4,tides,gr_potential; incl. host sync) %9.1f us   %6.2f TB/s (48 B)" % (t * 1e6, 48 * n / t / 1e12))
del dp

# prefix-sum path: star + 8 planets scattered among massless dust
m = min(n, 1_000_000)
cols, pp, _ = synthetic.disk(m, 8)
perm = np.concatenate([[0], 1 + np.random.default_rng(1).permutation(m - 1)])
cols = {k: np.ascontiguousarray(v[perm]) for k, v in cols.items()}
pp = {k: np.ascontiguousarray(v[perm]) for k, v in pp.items()}
host = dict(cols)
for k in ("tau_a", "tau_e", "tau_inc"):
    host[k] = pp[k]
dp = DeviceParticles.from_host(host)
view = dp.view()
for name, stack in (("gr", [("gr", {"c": C})]), ("modify_orbits_forces jacobi", [("modify_orbits_forces", {})])):
    o = build_ops(stack, pp)
    t = timed(lambda: ctx.apply(view, o, 0))
    print("  prefix-sum path, %-28s n = %d, 9 massive scattered: %9.1f us   %.3e evals/s" % (name, m, t * 1e6, m / t))
t = timed(lambda: ctx.gr_hamiltonian(view, C), reps=5)
print("  gr Hamiltonian (same particles; incl. host sync)          %9.1f us" % (t * 1e6))
