# dev: the last block's tail on the generated residues' row tiles only: tests + same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04l_gputest.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for P in fp32 f16; do for W in cfg4 cfg2; do for F in 0 1 0 1; do
  PF_SKIP_CONTEXT_ROWS=$F timeout 300 python bench.py --workload $W --precision $P $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P skip_context_rows=$F', round(d['ms_per_step'],4))"
done; done; done > gpurun_out/r04l_ab.txt 2>&1
cat gpurun_out/r04l_gputest.log gpurun_out/r04l_ab.txt
