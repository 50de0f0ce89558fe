#!/bin/bash
# dev: fresh-process shard check on two library builds (control: s_nop 0 in the LDS-DMA asm; fixed: s_nop 4)
mkdir -p gpurun_out
out=gpurun_out/r05a_race.txt; : > $out
run() { # label libpath n
  local bad=0
  for i in $(seq 1 $3); do
    if [ -n "$2" ]; then r=$(PF_LIB_PATH=$2 timeout 300 python tools/shard_check.py 64 128 3 2 2>&1 | tail -3); else r=$(timeout 300 python tools/shard_check.py 64 128 3 2 2>&1 | tail -3); fi
    echo "$1 proc $i: $(echo "$r" | tr '\n' ' ')" >> $out
    echo "$r" | grep -q "mismatches 0" || bad=$((bad+1))
  done
  echo "$1: $bad of $3 processes mismatched" | tee -a $out
}
t0=$(date +%s)
run nop0 $PWD/pepflowww_amd/lib/variants/libpf_nop0.so ${N0:-14}
t1=$(date +%s); echo "nop0 loop: $((t1-t0)) s" | tee -a $out
run nop4 "" ${N4:-24}
t2=$(date +%s); echo "nop4 loop: $((t2-t1)) s" | tee -a $out
