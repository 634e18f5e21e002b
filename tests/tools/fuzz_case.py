"""Development aid (GPU box): the numbers behind one case of tests/test_gpu_fuzz.py.
python tests/tools/fuzz_case.py SEED MODE"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as tf
from gpu_common import TOL, run_device, run_oracle
from reboundx_amd.device import Context
from test_gpu_parity import exact_and_sequential, is_prefix_sum_force

seed, mode = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(0)
cols, pp, stack, n, scan = tf.random_case(seed)
print("n", n, "scan", scan, "stack", stack)
ref = run_oracle(cols, pp, stack)
base = np.stack([cols["ax"], cols["ay"], cols["az"]])
got, _ = run_device(ctx, cols, pp, stack, 1.0, mode)
err = np.abs(got - ref)
upd = np.abs(ref - base).max(axis=0)
m = cols["m"]
total = float((m * upd).sum())
react = np.where(m > 0, (total - m * upd) / np.where(m > 0, m, 1.0), 0.0)
floor = np.abs(ref).max(axis=0) * (4 + np.sqrt(n)) * np.finfo(float).eps
worst = int((err.max(axis=0) / np.maximum(upd + react, floor / TOL)).argmax())
for i in sorted({0, 1, worst}):
    print("particle", i, "m", m[i], "err", err[:, i], "upd", upd[i], "react", react[i], "floor/TOL", floor[i] / TOL, "|ref|", np.abs(ref[:, i]).max(), "|base|", np.abs(base[:, i]).max())
    for f in stack:
        if is_prefix_sum_force(f):
            print("   checker - exact of", f[0], f[1], exact_and_sequential(cols, pp, f)[2][:, i])
    for k in range(len(stack)):
        one = run_oracle(cols, pp, [stack[k]])
        dev, _ = run_device(ctx, cols, pp, [stack[k]], 1.0, mode)
        print("   alone", stack[k][0], stack[k][1], "update", np.abs(one - base)[:, i].max(), "device - checker", np.abs(dev - one)[:, i])
from gpu_common import err_vs_exact
from test_gpu_parity import zero_acc
zc = zero_acc(cols)
for f in stack:
    if not is_prefix_sum_force(f):
        continue
    hi, lo, seq_err = exact_and_sequential(zc, pp, f)
    ctx.sequential_exact(False)
    dev, _ = run_device(ctx, zc, pp, [f], 1.0, mode)
    ctx.sequential_exact(True)
    dev_err = err_vs_exact(dev, hi, lo)
    body = np.abs(hi).max(axis=0)
    rel = (dev_err / np.where(body == 0, 1, body)[None, :]).max(axis=0)
    order = np.argsort(-rel)[:5]
    print("force", f, "massive bodies at", np.nonzero(m)[0][:20])
    for i in order:
        print("  body", i, "m", m[i], "rel", rel[i], "exact", hi[:, i], "device - exact", dev_err[:, i], "checker - exact", seq_err[:, i],
              "taus", [pp[k][i] for k in ("tau_a", "tau_e", "tau_inc")], "v", [cols[k][i] for k in ("vx", "vy", "vz")])
