# dev: f16-mode projection inside the score kernel: tests + same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -12 > gpurun_out/r04e_round4_tests.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call --precision f16"
for W in cfg4 cfg2; do for F in 0 1 0 1; do
  PF_FUSED_PROJ=$F timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W f16 fused_proj=$F', round(d['ms_per_step'],4), d['config']['launches_per_step'], (d.get('clocks_under_load') or {}).get('sclk_mhz'))"
done; done > gpurun_out/r04e_ab.txt 2>&1
cat gpurun_out/r04e_round4_tests.log gpurun_out/r04e_ab.txt
