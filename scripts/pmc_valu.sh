#!/bin/bash
# VALU / wait counters of the streaming kernel (one PMC pass; run on the GPU box from the repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/prof_valu" -o valu -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --grfull-n 0 --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_valu.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/prof_valu/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "stream_kernel" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("  %-22s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
