"""Runs the random-stack test of tests/test_gpu_fuzz.py on seeds beyond the ones pytest collects
(development aid; GPU box):  python tests/tools/fuzz_more.py FIRST COUNT [stack | integrators | steps]
  stack        the random stack, layout and parameters through rebxamd_apply (default)
  integrators  the random stack through integrate_force (every scheme)
  steps        the random stack through rebxamd_wh_steps"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as tf
from reboundx_amd.device import Context

first, count = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "stack"
test = {"stack": tf.run_case, "integrators": tf.test_random_stack_through_the_force_integrators,
        "steps": tf.test_random_stack_through_wisdom_holman_steps}[which]
ctx = Context(0)
bad = widened = 0
for seed in range(first, first + count):
    for mode in (1, 0):
        try:
            widened += 1 if test(ctx, seed, mode) else 0
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, "mode", mode, str(e)[:300], flush=True)
    if (seed - first) % 100 == 99:
        print("done", seed - first + 1, "failures", bad, flush=True)
print("total failures", bad, "| cases whose bar needed the binary128 term:", widened)
sys.exit(1 if bad else 0)
