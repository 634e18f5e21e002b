#!/bin/bash
# GPU time of the chunked Jacobi experiment (scripts/jacobi_chunked.py) from kernel traces, one run per chunk size.
# usage (GPU box, repository root): bash scripts/jacobi_chunked.sh [n]
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
N=${1:-10000000}
REPS=5
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 "$ROOT/scripts/jacobi_chunked.py" $N $REPS
for c in 0 524288 1048576 2097152; do
  rm -rf "$OUT/prof_jc_$c"
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_jc_$c" -o jc -- python3 "$ROOT/scripts/jacobi_chunked.py" $N $REPS $c > "$OUT/prof_jc_$c.log" 2>&1
  python3 - "$OUT/prof_jc_$c" $c $REPS <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
evals = 2 * int(sys.argv[3]) + 4
tot = 0.
parts = []
for r in csv.DictReader(open(f)):
    if "scan_" in r["Name"]:
        tot += float(r["TotalDurationNs"])
        parts.append("%s %d x %.1f us" % (r["Name"].split("(")[0][-28:], int(r["Calls"]) // evals, float(r["AverageNs"]) / 1e3))
print("chunk %s: %.1f us of kernels per evaluation | %s" % (sys.argv[2], tot / evals / 1e3, "; ".join(parts)))
P
done
