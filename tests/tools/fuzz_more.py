"""Runs the random-stack test of tests/test_gpu_fuzz.py on seeds beyond the ones pytest collects
(development aid; GPU box):  python tests/tools/fuzz_more.py FIRST COUNT"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as tf
from reboundx_amd.device import Context

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(0)
bad = 0
for seed in range(first, first + count):
    for mode in (1, 0):
        try:
            tf.test_random_stack_layout_and_parameters(ctx, seed, mode)
        except AssertionError as e:
            bad += 1
            print("FAIL seed", seed, "mode", mode, str(e)[:300], flush=True)
    if (seed - first) % 100 == 99:
        print("done", seed - first + 1, "failures", bad, flush=True)
print("total failures", bad)
sys.exit(1 if bad else 0)
