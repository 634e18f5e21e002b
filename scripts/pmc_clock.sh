#!/bin/bash
# Effective clock and VALU issue of the dominant kernels (one PMC pass; GPU box, repository root):
# clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (valid for dispatches of a millisecond or more).
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/prof_clock" -o clk -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_clock.log" 2>&1
python3 "$ROOT/scripts/summarize_clock.py" "$OUT"
