#!/bin/bash
# One pass over everything profiles/rNN/ holds for a commit (run ON the GPU box through gpurun; ~12 min):
#   tools/collect_profiles.sh <tag> <commit>
# -> gpurun_out/<tag>_*: rocprofv3 kernel statistics (cfg2 / cfg4, fp32 / f16), PMC HBM traffic, bench lines of un-profiled runs
TAG=${1:-r02}; COMMIT=${2:-unknown}; cd "$(dirname "$0")/.." && export TMPDIR=/tmp
bash tools/kernel_stats.sh $TAG fp32
bash tools/kernel_stats.sh $TAG f16
bash tools/pmc_traffic.sh $TAG $COMMIT
python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default.json
python bench.py --precision f16 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default_f16.json
python bench.py --workload cfg3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg3.json
python bench.py --workload cfg3 --precision f16 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg3_f16.json
python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cfg5.json
ls -la gpurun_out/${TAG}_* gpurun_out/pmc_traffic_${TAG}.json
