#!/bin/bash
# dev-only: kernel-trace of the training bench; where is the GPU idle between kernels?  (run ON the GPU box)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/gaps; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
# last 40 % of the trace = the timed steps
n0 = int(len(rows) * 0.6)
rows = rows[n0:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gap_after = collections.Counter(); cnt_after = collections.Counter()
end = rows[0][1]; prev = rows[0][2]
for s, e, n in rows[1:]:
    g = s - end
    if g > 0:
        gap_after[(prev, n)] += g; cnt_after[(prev, n)] += 1
    if e > end: end = e; prev = n
union = 0; ce = rows[0][0]
for s, e, _ in rows:
    if e > ce: union += e - max(s, ce); ce = e
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  sum of durations {busy/1e6:.2f} ms  union (any kernel running) {union/1e6:.2f} ms  idle {(span-union)/1e6:.2f} ms")
for (a, b), g in gap_after.most_common(25):
    print(f"{g/1e3:9.0f} us in {cnt_after[(a,b)]:5d} gaps (avg {g/cnt_after[(a,b)]/1e3:6.1f} us)  {a[:45]:45s} -> {b[:45]}")
PY
rm -rf $OUT
