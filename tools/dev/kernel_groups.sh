#!/bin/bash
# dev-only: kernel-trace of the training bench grouped by (kernel, grid size)  (run ON the GPU box)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/kgroups; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", "")))
rows.sort()
n = len(rows)
rows = rows[int(n * 2 / 3):]          # the last of the three replays
tot = collections.Counter(); cnt = collections.Counter()
for s, e, name, gx, gy, wx in rows:
    key = (name.replace("(anonymous namespace)::", "")[:48], gx, gy)
    tot[key] += e - s; cnt[key] += 1
print(f"one step: {len(rows)} kernels, busy {sum(tot.values())/1e6:.2f} ms")
for k, t in tot.most_common(45):
    print(f"{t/1e3:9.0f} us  {cnt[k]:4d} x {t/cnt[k]/1e3:8.1f} us  grid {k[1]:>9s} x {k[2]:>5s}  {k[0]}")
PY
rm -rf $OUT
