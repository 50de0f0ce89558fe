cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for F in 0 1 0 1; do
  PF_FUSED_PAIR=$F timeout 300 python bench.py --workload cfg2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 fp32 fused_pair=$F', round(d['ms_per_step'],4), d['config']['launches_per_step'])"
done > gpurun_out/r04p_ab.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_drift.py -x -q 2>&1 | tail -3 >> gpurun_out/r04p_ab.txt
cat gpurun_out/r04p_ab.txt
