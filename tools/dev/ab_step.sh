#!/bin/bash
# dev: same-box A/B of a plan choice inside the step:  tools/dev/ab_step.sh et_v5 [workload]   -> ms per step and the EdgeTransition launch, alternating
O=${1:-et_v5}; W=${2:-cfg4}
for r in 1 2; do for v in 0 1; do
  python bench.py --workload $W --no-modes --no-per-call --no-cpu-baseline --no-secondary --engine-opt $O=$v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$O=$v', 'ms_per_step %.4f' % d['ms_per_step'], 'ET launch %.1f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'sclk', d['roofline'].get('sclk_mhz'))"
done; done
