#!/bin/bash
# dev: bitwise repeat test of the projecting fp32 score kernel at B x L = 64 x 128 -- the form the engine runs (keys from the node state, KF)
# and the form before it -- N launches each, with the box's clock under load next to it
N=${N:-20000}
for i in 1 2 3; do
  r=$(PF_REPEAT_LAUNCHES=$N python -m pytest tests/test_gpu_fresh_process.py -q -k "many_launches and True" 2>&1 | grep -E "launches differ|passed" | tail -1)
  echo "KF form, $N launches, run $i: $r"
done
r=$(PF_REPEAT_LAUNCHES=$N python tools/dev/r05_repeat_old_form.py 2>&1 | tail -1)
echo "previous form, $N launches: $r"
/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2
