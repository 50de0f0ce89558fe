#!/bin/bash
# rocprofv3 kernel-trace stats of the bench (run ON the GPU box through gpurun); top rows -> gpurun_out/<tag>_<workload>_kernel_stats.csv
TAG=${1:-v3}; PREC=${2:-fp32}; cd "$(dirname "$0")/.." && export TMPDIR=/tmp
[ $PREC = fp32 ] || TAG=${TAG}_$PREC
for W in cfg2 cfg4 cfg3; do
  if [ $W = cfg2 ]; then K=20; WU=3; else K=10; WU=2; fi
  OUT=gpurun_out/ks_${TAG}_$W; rm -rf $OUT
  PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps $K --warmup $WU --no-graph --no-cpu-baseline --no-secondary --no-modes --no-per-call --workload $W --precision $PREC > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  (head -1 "$f"; grep "anonymous namespace" "$f" | head -24) | cut -c1-260 > gpurun_out/${TAG}_${W}_kernel_stats.csv
  rm -rf $OUT
  python bench.py --workload $W --precision $PREC --no-cpu-baseline --no-secondary --no-modes --no-per-call 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$W.json
done
