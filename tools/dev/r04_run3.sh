# dev: EdgeTransition v4 variants alone (B=64, L=128), two rounds each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do for v in old base early; do
  timeout 300 python tools/dev/et_bench.py v4 fp32 pepflowww_amd/lib/variants/libpf_$v.so 2>&1 | grep "us per launch"
done; done > gpurun_out/r04c_et_variants.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "edge_transition" 2>&1 | tail -2 >> gpurun_out/r04c_et_variants.txt
cp pepflowww_amd/lib/variants/libpf_early.so pepflowww_amd/lib/libpepflow_hip.so     # (scratch copy of the repo on the GPU box)
echo "early variant as the library:" >> gpurun_out/r04c_et_variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "edge_transition" 2>&1 | tail -2 >> gpurun_out/r04c_et_variants.txt
timeout 900 python -m pytest tests/test_gpu_bigshape.py -x -q 2>&1 | tail -2 >> gpurun_out/r04c_et_variants.txt
cat gpurun_out/r04c_et_variants.txt
