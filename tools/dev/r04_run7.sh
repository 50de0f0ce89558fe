# dev: same-box A/B of two builds of the library (PF_LIB_PATH): $1 = the other .so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OLD=$1
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -3 > gpurun_out/r04n_tests.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for P in fp32 f16; do for W in cfg4 cfg2; do for F in old new old new; do
  if [ $F = old ]; then export PF_LIB_PATH=$PWD/$OLD; else unset PF_LIB_PATH; fi
  timeout 300 python bench.py --workload $W --precision $P $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P lib=$F', round(d['ms_per_step'],4))"
done; done; done > gpurun_out/r04n_ab.txt 2>&1
cat gpurun_out/r04n_tests.log gpurun_out/r04n_ab.txt
