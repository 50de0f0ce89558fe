#!/bin/bash
# dev: same-box A/B of a variant library (tools/dev/build_variant.sh) against the shipped one inside the step, alternating:
#   tools/dev/ab_lib.sh <variant> <workload> [bench flags]
V=$1; W=$2; shift 2
for r in 1 2; do for lib in $V new; do
  if [ $lib = new ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$V.so; fi
  python bench.py --workload $W --no-modes --no-per-call --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib $W $*', 'ms_per_step %.4f' % d['ms_per_step'], 'ET launch %.1f us' % d['roofline']['avg_launch_us'])"
done; done
