#!/bin/bash
# HBM traffic of every kernel of the all-massive stack (PARTICLE and Jacobi coordinates), from the TCC counters in two separate
# passes (FETCH_SIZE, WRITE_SIZE), per body.  GPU box, repository root.  usage: bash scripts/pmc_c3_traffic.sh [n]
# Units as in scripts/summarize_profiles.py: both counters in KB; FETCH_SIZE doubled on gfx950 (128-byte requests of wide
# coalesced loads are tallied at 64 bytes), WRITE_SIZE as it is.
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
N=${1:-1e7}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_c3_fetch" -o f -- python3 "$ROOT/scripts/bench_c3.py" $N 4 > "$OUT/prof_c3_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_c3_write" -o w -- python3 "$ROOT/scripts/bench_c3.py" $N 4 > "$OUT/prof_c3_write.log" 2>&1
python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], float(sys.argv[2])
def mean_by_kernel(d, counter):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "rebxamd::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe, wr = mean_by_kernel("prof_c3_fetch", "FETCH_SIZE"), mean_by_kernel("prof_c3_write", "WRITE_SIZE")
print("%-72s %6s %12s %12s %10s" % ("kernel (mean per launch, %g bodies)" % n, "calls", "read B/body", "write B/body", "sum"))
for k in sorted(fe, key=lambda k: -fe[k][0]):
    r = 2 * fe[k][0] * 1024 / n
    w = wr.get(k, (0., 0))[0] * 1024 / n
    if r + w > 0.5:
        print("%-72s %6d %12.1f %12.1f %10.1f" % (k, fe[k][1], r, w, r + w))
PY
