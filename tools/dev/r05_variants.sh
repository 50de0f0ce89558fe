#!/bin/bash
# dev: fresh-process shard check over several library builds:  tools/dev/r05_variants.sh OUTTAG N name1 name2 ...   ("main" = the in-tree lib)
mkdir -p gpurun_out
tag=$1; n=$2; shift 2
out=gpurun_out/${tag}.txt; : > $out
for name in "$@"; do
  bad=0; t0=$(date +%s)
  for i in $(seq 1 $n); do
    if [ "$name" = main ]; then r=$(timeout 300 python tools/shard_check.py ${SHAPE:-64 128 3 2} 2>&1 | grep -v amdgpu.ids | tail -2)
    else r=$(PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$name.so timeout 300 python tools/shard_check.py ${SHAPE:-64 128 3 2} 2>&1 | grep -v amdgpu.ids | tail -2); fi
    echo "$name proc $i: $(echo "$r" | tr '\n' ' ')" >> $out
    echo "$r" | grep -q "mismatches 0" || bad=$((bad+1))
  done
  echo "$name: $bad of $n processes mismatched ($(( $(date +%s) - t0 )) s)" | tee -a $out
done
