#!/bin/bash
# dev-only: kernel-trace of the training bench (eager, so that every launch is traced), ONE step cut out between two launches of the
# loss kernel, grouped by kernel name  (run ON the GPU box)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=/tmp/kgroups; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-per-call --no-graph "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "train_losses_kernel" in r[2] or "train_loss" in r[2].lower() and "bwd" not in r[2].lower()]
if len(marks) < 2:
    marks = [i for i, r in enumerate(rows) if "loss" in r[2].lower()]
a, b = marks[-2], marks[-1]
step = rows[a:b]
wall = (step[-1][1] - step[0][0]) / 1e6
tot = collections.Counter(); cnt = collections.Counter()
for s, e, name in step:
    key = re.sub(r"\(anonymous namespace\)::", "", name)
    key = re.sub(r"^void ", "", key)[:70]
    tot[key] += e - s; cnt[key] += 1
print(f"one step: {len(step)} kernels, busy {sum(tot.values())/1e6:.2f} ms, wall {wall:.2f} ms")
for k, t in tot.most_common(int(__import__("os").environ.get("KG_TOP", "60"))):
    print(f"{t/1e3:9.0f} us  {cnt[k]:4d} x {t/cnt[k]/1e3:8.1f} us  {k}")
PY
rm -rf $OUT
