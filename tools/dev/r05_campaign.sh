#!/bin/bash
# dev: the fresh-process shard check of the SHIPPED library over the shapes the projection-inside form runs at, both precision modes
mkdir -p gpurun_out; out=gpurun_out/r05_campaign.txt; : > $out
for prec in fp32 f16; do
  for shape in "16 64" "64 64" "64 80" "64 96" "64 112" "64 128"; do
    bad=0; n=${N:-20}
    for i in $(seq 1 $n); do
      r=$(timeout 300 python tools/shard_check.py $shape 3 2 $prec 2>&1 | grep -v amdgpu.ids | tail -1)
      echo "$r" | grep -q "mismatches 0" || { bad=$((bad+1)); echo "$prec $shape proc $i: $r" >> $out; }
    done
    echo "$prec B x L = $shape: $bad of $n fresh processes mismatched" | tee -a $out
  done
done
