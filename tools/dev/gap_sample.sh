#!/bin/bash
# dev-only: kernel trace of the sampling bench IN GRAPH MODE: busy vs idle per step, gaps by predecessor  (run ON the GPU box)
W=${1:-cfg2}; cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/gaps_s; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]))
rows.sort()
# find the last 10 occurrences of sampler_step (step ends) and analyse the span between them
ends = [i for i, r in enumerate(rows) if "sampler_step" in r[2]]
i0, i1 = ends[-11], ends[-1]
seg = rows[i0 + 1:i1 + 1]
span = seg[-1][1] - rows[i0][1]
busy = sum(e - s for s, e, _ in seg)
print(f"10 steps: {len(seg)} kernels, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us  -> per step {span/1e4:.1f} us, idle {(span-busy)/1e4:.1f} us")
dur = collections.Counter(); cnt = collections.Counter(); gap = collections.Counter()
prev_end = rows[i0][1]; prev = rows[i0][2]
for s, e, n in seg:
    dur[n] += e - s; cnt[n] += 1
    gap[n] += max(0, s - prev_end)         # idle time right BEFORE kernel n
    prev_end = max(prev_end, e)
for n, t in dur.most_common(20):
    print(f"{n:40s} {cnt[n]/10:5.1f}/step  avg {t/cnt[n]/1e3:7.2f} us  gap before avg {gap[n]/cnt[n]/1e3:5.2f} us  -> {t/1e4:6.1f} + {gap[n]/1e4:5.1f} us/step")
PY
rm -rf $OUT
