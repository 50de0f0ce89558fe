# dev: helper waves of the projection prologue (L <= 64): tests + same-box A/B at cfg2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_drift.py -x -q 2>&1 | tail -3 > gpurun_out/r04o_tests.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for P in fp32 f16; do for F in 0 1 0 1; do
  PF_PROJ_HELPERS=$F timeout 300 python bench.py --workload cfg2 --precision $P $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 $P helpers=$F', round(d['ms_per_step'],4))"
done; done > gpurun_out/r04o_ab.txt 2>&1
cat gpurun_out/r04o_tests.log gpurun_out/r04o_ab.txt
