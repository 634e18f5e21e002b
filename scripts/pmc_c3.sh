#!/bin/bash
# Counters of the prefix-sum passes in the all-massive Jacobi stack (one PMC pass; GPU box, repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/prof_c3pmc" -o c3 -- python3 "$ROOT/scripts/bench_c3.py" ${1:-1e7} 4 > "$OUT/prof_c3pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/prof_c3pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rebxamd::scan" in k:
        acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m["SQ_WAVES"]
    print("%-50s waves %7.0f  per wave: VALU %6.0f SALU %5.0f LDS %5.0f  wave_cycles %7.0f wait_any %7.0f wait_lds %6.0f" % (
        k, w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m["SQ_WAVE_CYCLES"] / w, m["SQ_WAIT_ANY"] / w, m["SQ_WAIT_INST_LDS"] / w))
PY
