#!/bin/bash
# dev: rocprofv3 kernel statistics of ONE bench workload -> gpurun_out/<tag>_<workload>_kernel_stats.csv   (tools/dev/ks_one.sh tag workload [bench flags])
TAG=$1; W=$2; shift 2; cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
OUT=gpurun_out/ks_${TAG}_$W; rm -rf $OUT
PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-secondary --no-modes --no-per-call --workload $W "$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
(head -1 "$f"; sed -n 2,60p "$f") | cut -c1-220 > gpurun_out/${TAG}_${W}_kernel_stats.csv
rm -rf $OUT
