#!/bin/bash
# dev-only (run ON the GPU box): per-launch durations of the row-sized GEMM kernels of ONE eager training step, with their grids
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=/tmp/dtrace; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-per-call --no-graph "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "train_losses_kernel" in r[2]]
a, b = marks[-2], marks[-1]
agg = collections.defaultdict(list)
for s, e, name, gx, wx in rows[a:b]:
    if "gemm_f32" in name:
        key = (re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:40], gx)
        agg[key].append((e - s) / 1e3)
for (name, gx), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:40s} grid {gx:>8s}  n {len(v):3d}  avg {sum(v)/len(v):7.1f} us  total {sum(v):8.0f} us")
PY
rm -rf $OUT
