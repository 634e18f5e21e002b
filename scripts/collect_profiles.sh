#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box, from
# the repository root):  per-kernel timing of the default bench command, and HBM traffic of the
# streaming kernel from PMC counters in two separate passes (FETCH_SIZE, WRITE_SIZE).
# Raw outputs land in gpurun_out/prof_*; scripts/summarize_profiles.py condenses them into profiles/.
set -e
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
# (the out-of-cache section launches the same kernel instance on 4e7 grains: it is traced on its own below, so that the
# average of the kernel's row in the first summary is the headline size alone)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace" -o bench -- python3 "$ROOT/bench.py" --no-cpu --ooc-n 0 > "$OUT/prof_trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --grfull-n 0 --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -o write -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --grfull-n 0 --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 > "$OUT/prof_write.log" 2>&1
# the same kernel out of cache: 4e7 grains (the bench's out-of-cache section; the headline part of this run is shrunk to nothing)
OOC="--n-per-gpu 1000 --steps 1 --warmup 0 --no-cpu --grfull-n 0 --c5-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 40000000"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace_ooc" -o ooc -- python3 "$ROOT/bench.py" $OOC > "$OUT/prof_trace_ooc.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch_ooc" -o fetch -- python3 "$ROOT/bench.py" $OOC > "$OUT/prof_fetch_ooc.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write_ooc" -o write -- python3 "$ROOT/bench.py" $OOC > "$OUT/prof_write_ooc.log" 2>&1
find "$OUT"/prof_trace "$OUT"/prof_fetch "$OUT"/prof_write -name "*.csv" | head -20
