"""Development aid (GPU box): reference-order arithmetic on layouts with massive bodies anywhere --
ulp distance of gr and Jacobi/barycentric modify_orbits_forces from the CPU checker, and the time
of the sequential-order route at its size limit.   python tests/tools/seq_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from gpu_common import run_device, run_oracle, ulp_diff
from reboundx_amd import synthetic
from reboundx_amd.device import Context
from test_gpu_parity import SCAN_FORCES, scatter_planets, zero_acc

ctx = Context(0)
bad = 0
for label, n, planets, dust, scatter, kw in (("all-massive", 70, 2, 1e-9, False, {}), ("all-massive", 257, 2, 1e-9, False, {}),
                                             ("all-massive", 6000, 2, 1e-9, False, {}), ("scattered", 9001, 6, 0., True, {}),
                                             ("long-prefix", 4000, 99, 0., False, dict(n_head=100, tail_massless=True)),
                                             ("all-massive", 30000, 2, 1e-9, False, {})):
    cols, pp, _ = synthetic.disk(n, planets, dust_mass=dust) if dust else synthetic.disk(n, planets)
    if scatter:
        cols, pp = scatter_planets(cols, pp)
    for newton in (False, True):
        c = zero_acc(cols)
        if newton:
            synthetic.add_newtonian(c, n_head=min(n, 8))
        for force in SCAN_FORCES:
            if n > 10000 and force[0] == "gr":
                continue
            ref = run_oracle(c, pp, [force])
            got, _ = run_device(ctx, c, pp, [force], 1.0, 1, **kw)
            u = ulp_diff(got, ref)
            print(label, n, "newton" if newton else "zero", force[0], force[1], "max ulp", int(u.max()), "bodies off", int((u.max(axis=0) > 0).sum()), flush=True)
            bad += int(u.max() > 0)
# random prefixes of massive and massless bodies followed by massless grains: the chains stop at the last massive body
rng = np.random.default_rng(11)
for case in range(int(os.environ.get("SEQ_TAIL_CASES", "120"))):
    prefix = int(rng.choice([66, 255, 256, 257, 300, 1025, 2500]))
    tail = int(rng.choice([1, 2, 3, 255, 256, 1000, 20000, 70000]))
    n = prefix + tail
    cols, pp, _ = synthetic.disk(n, 2, seed=int(rng.integers(1, 100)))
    massive = int(rng.integers(3, prefix - 3))
    cols["m"][rng.choice(np.arange(3, prefix - 1), size=massive, replace=False)] = 10.0 ** rng.uniform(-10, -4, massive)
    cols["m"][prefix - 1] = 10.0 ** rng.uniform(-10, -4)
    cols["m"][prefix:] = 0.
    if rng.random() < 0.6:
        synthetic.add_newtonian(cols, n_head=prefix)
    else:
        cols = zero_acc(cols)
    k = int(rng.integers(0, len(SCAN_FORCES)))
    stack = [SCAN_FORCES[k]]
    if rng.random() < 0.4:
        stack.append(SCAN_FORCES[2 if k < 2 else 0])        # gr and a centre-of-mass drag one after the other
    ref = run_oracle(cols, pp, stack)
    got, _ = run_device(ctx, cols, pp, stack, 1.0, 1)
    u = ulp_diff(got, ref)
    if u.max() > 0:
        bad += 1
        print("tail case", case, "prefix", prefix, "tail", tail, "massive", massive, [f[0] + str(f[1]) for f in stack], "max ulp", int(u.max()),
              "bodies off", int((u.max(axis=0) > 0).sum()), "first", int(np.nonzero(u.max(axis=0))[0][0]), flush=True)
print("tail cases done, off so far:", bad, flush=True)
for n in (65536,):
    cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-9)
    for force in SCAN_FORCES:
        for seq in (True, False):
            ctx.sequential_exact(seq)
            run_device(ctx, zero_acc(cols), pp, [force], 1.0, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_device(ctx, zero_acc(cols), pp, [force], 1.0, 1)
            torch.cuda.synchronize()
            print("n", n, force[0], force[1], "sequential" if seq else "scans", "%.2f ms (incl. upload)" % ((time.perf_counter() - t0) * 1e3), flush=True)
    ctx.sequential_exact(True)
print("cases off:", bad)
sys.exit(1 if bad else 0)
