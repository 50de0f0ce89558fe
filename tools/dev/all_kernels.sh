#!/bin/bash
# dev-only: per-kernel average durations of one denoise step (graph off), all of our kernels  (run ON the GPU box)
W=${1:-cfg2}; cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/allk; rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 20 --warmup 3 --no-graph --no-cpu-baseline --workload $W > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
    calls = int(r["Calls"]); per_step = calls / 34.0
    avg = float(r["AverageNs"]) / 1e3
    print(f"{r['Name'][:64]:64s} calls {calls:5d} ({per_step:5.1f}/step) avg {avg:8.1f} us  -> {per_step*avg:7.1f} us/step")
    tot += per_step * avg
print("sum of our kernels per step:", round(tot, 1), "us")
PY
rm -rf $OUT
