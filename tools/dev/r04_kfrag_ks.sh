# dev: per-kernel averages (rocprofv3, eager) and graph-mode step time of cfg3 fp32 with PF_K_FRAG=0 / 1, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for F in 0 1 0 1; do
  export PF_K_FRAG=$F
  timeout 300 python bench.py --workload cfg3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 fp32 graph k_frag=$F', round(d['ms_per_step'],4))"
done
for F in 0 1; do
  export PF_K_FRAG=$F
  OUT=/tmp/ks_kf$F; rm -rf $OUT
  PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 10 --warmup 2 --no-graph $B --workload cfg3 > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "k_frag=$F"; grep "anonymous namespace" "$f" | head -7 | python -c "
import sys, csv
for r in csv.reader(sys.stdin): print('   ', r[0][:70].replace('void (anonymous namespace)::',''), r[1], round(float(r[3])/1000,1))"
done
