#!/bin/bash
# A/B of pf_ipa_attn_args.fused_pair in the step (run ON the GPU box): step time of both forms + rocprofv3 averages of the attention kernels
mkdir -p gpurun_out/r03f; export TMPDIR=/tmp
for P in fp32 f16; do for W in ${WL:-cfg4 cfg2 cfg3}; do for F in 0 1; do
  PF_FUSED_PAIR=$F python bench.py --workload $W --precision $P --no-cpu-baseline --no-secondary --no-modes --no-per-call --steps 40 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03f/bench_${W}_${P}_fused$F.json
  python -c "import json; d=json.load(open('gpurun_out/r03f/bench_${W}_${P}_fused$F.json')); print('$W $P fused=$F ms_per_step', round(d['ms_per_step'],4))"
done; done; done
for P in fp32 f16; do for F in 0 1; do
  OUT=/tmp/ks_$P$F; rm -rf $OUT
  PF_FUSED_PAIR=$F rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 2 --no-graph --no-cpu-baseline --no-secondary --no-modes --no-per-call --workload cfg4 --precision $P > /dev/null 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== cfg4 $P fused=$F"; grep -E "ipa_scores|ipa_pair" "$f" | cut -d, -f1-4 | cut -c1-160
  rm -rf $OUT
done; done
