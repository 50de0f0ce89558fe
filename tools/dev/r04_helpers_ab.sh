# dev: same-box A/B of the prologue's helper waves (L <= 64, fp32 form): PF_PROJ_HELPERS=0 / 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for F in 0 1 0 1 0 1; do
  PF_PROJ_HELPERS=$F timeout 300 python bench.py --workload cfg2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 helpers=$F', round(d['ms_per_step'],4))"
done
