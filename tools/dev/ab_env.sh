#!/bin/bash
# A/B of an environment switch in the step (run ON the GPU box): tools/dev/ab_env.sh VAR "v1 v2 ..." "workloads" "precisions"
VAR=$1; VALS=$2; WLS=${3:-cfg2}; PRECS=${4:-fp32}
for P in $PRECS; do for W in $WLS; do for V in $VALS; do
  env $VAR=$V python bench.py --workload $W --precision $P --no-cpu-baseline --no-secondary --no-modes --no-per-call --steps 60 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P $VAR=$V ms_per_step', round(d['ms_per_step'],4))"
done; done; done
