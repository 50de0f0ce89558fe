#!/bin/bash
# dev-only (run ON the GPU box): rebuild ONE source of the library with extra -D flags, print the bench per-kernel timings, restore.
#   tools/dev/exp_file.sh <source-stem> "<flags>" [workload]
R=$GRAFT_REPO_ROOT; cd $R
F=$1; FLAGS=$2; W=${3:-cfg4}
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FLAGS -c pepflowww_amd/csrc/$F.hip -o /tmp/expx.o || exit 1
objs=""; for f in pepflowww_amd/lib/*.o; do [ "$f" != "pepflowww_amd/lib/$F.o" ] && objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/expx.o
timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$F [$FLAGS] $W', round(d['ms_per_step'],4), {k: round(v['avg_launch_us'],1) for k,v in d.get('kernel_us_dbg', {}).items()} or (d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us']))"
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
