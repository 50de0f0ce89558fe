# dev: kernel statistics at HEAD (cfg4 / cfg2, fp32) + same-box A/B of the fp32 fused pair aggregation
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for W in cfg4 cfg2; do
  OUT=gpurun_out/ks_r04b_$W; rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 2 --no-graph $B --workload $W > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  (head -1 "$f"; grep "anonymous namespace" "$f" | head -24) | cut -c1-260 > gpurun_out/r04b_${W}_kernel_stats.csv
  rm -rf $OUT
done
for W in cfg4 cfg2; do for F in 0 1 0 1; do
  PF_FUSED_PAIR=$F timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W fused_pair=$F', round(d['ms_per_step'],4), d['config']['launches_per_step'], d.get('clocks_under_load'))"
done; done > gpurun_out/r04b_ab.txt 2>&1
cat gpurun_out/r04b_ab.txt
