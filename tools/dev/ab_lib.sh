#!/bin/bash
# same-box A/B of two libraries in the step
W=$1; shift
for r in 1 2; do for lib in old new; do
  if [ $lib = old ]; then export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_old.so; else unset PF_LIB_PATH; fi
  python bench.py --workload $W --no-modes --no-per-call --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib $W $*', 'ms_per_step %.4f' % d['ms_per_step'], 'ET launch %.1f us' % d['roofline']['avg_launch_us'])"
done; done
