#!/bin/bash
# dev: the shard check in N fresh processes (first-launch conditions every time)
N=${1:-30}
for i in $(seq 1 $N); do
  OUTER=1 python tools/dev/r04_det3.py 2>&1 | grep -v amdgpu.ids | grep -v "mismatches 0" | sed "s/^/[proc $i] /"
done
echo "fresh-process runs done: $N"
