# dev: the last EdgeTransition without its z' store: tests + same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -k "encoder or sampl or step or masked or edge_transition" 2>&1 | tail -4 > gpurun_out/r04k_tests.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for P in fp32 f16; do for W in cfg4 cfg2; do for F in 1 0 1 0; do
  PF_ET_LAST_STORE=$F timeout 300 python bench.py --workload $W --precision $P $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P last_store=$F', round(d['ms_per_step'],4))"
done; done; done > gpurun_out/r04k_ab.txt 2>&1
cat gpurun_out/r04k_tests.log gpurun_out/r04k_ab.txt
