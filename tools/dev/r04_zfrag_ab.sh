# dev: same-box A/B of the fragment-ordered pair tensor (PF_ET_ZFRAG=0 / 1), python bench.py --workload W, ms per step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for W in cfg4 cfg2 cfg3; do for F in 0 1 0 1; do
  PF_ET_ZFRAG=$F timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W z_frag=$F', round(d['ms_per_step'],4), 'ET', round(d['roofline']['avg_launch_us'],1))"
done; done
