#!/bin/bash
# dev: the shipped library against several variant libraries inside one workload's step, alternating:  tools/dev/ab_many.sh <workload> v1 v2 ...
W=$1; shift
for r in 1 2; do for lib in new "$@"; do
  if [ $lib = new ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$lib.so; fi
  python bench.py --workload $W --no-modes --no-per-call --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-12s' % '$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'ET launch %.1f us' % d['roofline']['avg_launch_us'])"
done; done
