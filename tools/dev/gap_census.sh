#!/bin/bash
# dev: idle time between the kernels of the graph-replayed step: rocprofv3 kernel trace (timestamps) of `bench.py --workload W` WITH the hipGraph,
# then per step: sum of kernel durations, wall from the first start to the last end, the gaps.   tools/dev/gap_census.sh [workload] [bench flags]
W=${1:-cfg4}; shift
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/gap_$W; rm -rf $OUT
PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-per-call --workload $W "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the timed region: the last 20 * n kernels whose pattern repeats -- take the last 10 steps by the EdgeTransition launches (5 per step)
et = [i for i, e in enumerate(ev) if "edge_transition" in e[2]]
per = 5
last = et[-per * 10:]
i0 = last[0]
# a step starts at the kernel after the previous step's last kernel: use the window from the first of these ET launches back to the previous ET + 1 ... simpler: window = [ET k, ET k + 50 launches)
seg = ev[et[-per * 10 - 1] + 1: et[-1] + 1]          # ten steps' worth, aligned just behind an ET launch
busy = sum(e - s for s, e, _ in seg)
wall = seg[-1][1] - seg[0][0]
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
print("kernels in window %d   wall %.1f us   busy %.1f us   idle %.1f us (%.1f %%)   per step: wall %.1f busy %.1f" % (len(seg), wall / 1e3, busy / 1e3, (wall - busy) / 1e3, 100 * (wall - busy) / wall, wall / 1e4, busy / 1e4))
g = sorted(gaps)
print("gap between consecutive kernels: median %.2f us, mean %.2f, p90 %.2f, max %.2f; negative (overlap) %d" % (g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3, g[int(0.9 * len(g))] / 1e3, g[-1] / 1e3, sum(x < 0 for x in g)))
by = collections.defaultdict(list)
for (s, e, n), gp in zip(seg[1:], gaps):
    by[n.split("(")[0][-48:]].append(gp)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("   gap in front of %-50s n %4d mean %.2f us" % (n, len(v), sum(v) / len(v) / 1e3))
PY
rm -rf $OUT
