"""Development aid (GPU box): reference-order arithmetic over the random stacks of tests/test_gpu_fuzz.py -- which bodies
do not carry the checker's bits, and how far off they are.   python tests/tools/fuzz_bits.py [first] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpu_common import run_device, run_oracle, ulp_diff
from reboundx_amd.device import Context
import test_gpu_fuzz as tf

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ctx = Context(0)
off_cases = 0
for seed in range(first, first + count):
    cols, pp, stack, n, scan = tf.random_case(seed)
    ref = run_oracle(cols, pp, stack)
    got, _ = run_device(ctx, cols, pp, stack, 1.0, 1)
    u = ulp_diff(got, ref).max(axis=0)
    bad = np.nonzero(u)[0]
    if len(bad):
        off_cases += 1
        massive = np.nonzero(cols["m"])[0]
        print("seed", seed, "n", n, "scan", scan, [f[0] + str(f[1].get("coordinates", "")) for f in stack], "bodies off", len(bad), "first", bad[:6].tolist(),
              "max ulp", int(u.max()), "massive", len(massive), "off bodies massive:", bool(np.isin(bad, massive).all()), flush=True)
print("cases with bodies off:", off_cases, "of", count)
