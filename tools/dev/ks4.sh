#!/bin/bash
# dev-only (run ON the GPU box): rocprofv3 kernel averages of the cfg4 bench, one compact line per kernel.  ks4.sh [fp32|f16]
PREC=${1:-fp32}; cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=/tmp/ks4_$$; rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-secondary --workload ${W:-cfg4} --precision $PREC > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)[:60]
    print(f'{name:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f} %')
PY
python bench.py --workload ${W:-cfg4} --precision $PREC --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4))"
rm -rf $OUT
