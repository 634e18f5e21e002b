#!/bin/bash
# VALU counters of the device-resident stepping kernels (one PMC pass; GPU box, repository root).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/prof_valu2" -o valu -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --grfull-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 --c5-steps 50 > "$OUT/prof_valu2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/prof_valu2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rebxamd" in k:
        acc[k[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k, "n=%d" % len(d["SQ_WAVES"]))
    print("   waves %.0f  VALU/wave %.0f  SALU/wave %.0f  VMEM_RD/wave %.1f  wave_cycles/wave %.0f" % (
        w, sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"]) / w, sum(d["SQ_INSTS_SALU"]) / len(d["SQ_INSTS_SALU"]) / w,
        sum(d["SQ_INSTS_VMEM_RD"]) / len(d["SQ_INSTS_VMEM_RD"]) / w, sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"]) / w))
PY
