set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -15 > gpurun_out/r04a_round4_tests.log
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for W in cfg4 cfg2; do
for F in 0 1 0 1; do
  PF_FUSED_PROJ=$F timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 > gpurun_out/r04a_bench_${W}_proj$F.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04a_bench_${W}_proj$F.json"))
print("$W proj=$F", d["ms_per_step"], d.get("roofline",{}).get("achieved"), d.get("launches_per_step"))
PY
done
done > gpurun_out/r04a_ab.txt 2>&1
cat gpurun_out/r04a_round4_tests.log gpurun_out/r04a_ab.txt
