#!/bin/bash
# Drop-in call (host array-of-structs in and out) at several particle counts (GPU box, repository root).
set -e
for n in 1000 10000 30000 100000 300000 1000000 3000000 10000000; do
  timeout -k 10 120 python3 scripts/bench_dropin.py $n gr 2>&1 | tail -2
done
