# dev: same-box A/B of the k fragments of the projection-launch path (fp32 mode, PF_K_FRAG=0 / 1): python bench.py --workload cfg3, ms per step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for F in 0 1 0 1; do
  PF_K_FRAG=$F timeout 300 python bench.py --workload cfg3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 fp32 k_frag=$F', round(d['ms_per_step'],4))"
done
