"""profiles/rNN_parity_report.txt from gpurun_out/parity_report.jsonl (what the -m gpu parity tests recorded in their last
full run; the largest figure per case).  usage: python scripts/parity_report.py [round]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = {}
for line in open(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")):
    r = json.loads(line)
    c = rows.setdefault(r.pop("case"), {})
    for k, v in r.items():
        c[k] = max(c.get(k, 0.0), v)
head = """# Error figures recorded by the -m gpu parity tests (tests/gpu_common.record), last full run of the suite.
# floored: relative to max(|reference component|, 1e-3 of the body's largest component); unfloored: relative to the component itself
# (components that vanish by symmetry make it large without meaning anything).  scan:* cases compare against the SEQUENTIAL checker,
# whose own distance from the binary128 value is what these figures show (exact:* cases give the device's distance from that value).
"""
fmt = lambda c, k: ("%10.2e" % c[k]) if k in c else "%10s" % "-"
with open(os.path.join(ROOT, "profiles", "r%02d_parity_report.txt" % rnd), "w") as f:
    f.write(head)
    f.write("%-76s %10s %10s  other\n" % ("case", "floored", "unfloored"))
    for case in sorted(rows):
        c = rows[case]
        other = " ".join("%s=%.2e" % (k, v) for k, v in sorted(c.items()) if k not in ("floored", "unfloored"))
        f.write("%-76s %s %s  %s\n" % (case, fmt(c, "floored"), fmt(c, "unfloored"), other))
print(len(rows), "cases")
