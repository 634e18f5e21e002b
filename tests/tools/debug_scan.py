import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from gpu_common import run_device, run_oracle
from reboundx_amd import synthetic
from reboundx_amd.device import Context
ctx = Context(0)
n = 6000
cols, pp, _ = synthetic.disk(n, 2, dust_mass=1e-9)
for k in ("ax", "ay", "az"): cols[k] = np.zeros(n)
f = [("gr", {"c": synthetic.C_DEFAULT})]
ref = run_oracle(cols, pp, f)
for mode in (1, 0):
    got, _ = run_device(ctx, cols, pp, f, 1.0, mode)
    d = np.abs(got - ref)
    big = np.abs(ref).max(axis=0)
    e = (d / big).max(axis=0)
    idx = np.argsort(e)[-8:]
    print("mode", mode, "max err rel-to-largest-comp", e.max())
    for i in idx:
        print(i, e[i], ref[:, i], got[:, i] - ref[:, i])
    # error vs index
    for a in range(0, n, 500):
        print(a, e[a:a + 500].max())
