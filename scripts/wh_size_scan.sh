#!/bin/bash
# Queued and host-synced Wisdom-Holman step times of the resident run at several sizes.
set -e
mkdir -p gpurun_out
for n in 250000 1000000 4000000 10000000; do
  timeout -k 10 200 python3 bench.py --no-cpu --grfull-n 0 --c3-n 0 --c3-big-n 0 --ooc-n 0 --steps 3 --warmup 1 --c5-n $n --c5-steps 2000 > gpurun_out/wh_scan_$n.json
  python3 - $n <<'P'
import json, sys
d = json.load(open("gpurun_out/wh_scan_%s.json" % sys.argv[1]))["resident_run"]
print(d["n"], "synced %.1f us (open run %.1f us)  queued %.1f us  frac %.3f | star+planet queued %.1f us" % (d["us_per_step_host_synced"], d["us_per_step_host_synced_open_run"], d["us_per_step_queued"], d["roofline"]["frac"], d["star_planet_grains"]["us_per_step_queued"]))
P
done
