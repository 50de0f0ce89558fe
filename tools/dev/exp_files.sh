#!/bin/bash
# dev-only (run ON the GPU box): rebuild SEVERAL sources with extra -D flags, print step times, restore.
#   tools/dev/exp_files.sh "<stem stem ...>" "<flags>" "<workloads>"
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
objs=""; for f in pepflowww_amd/lib/*.o; do objs="$objs $f"; done
for F in $1; do
  X=""; [ "$F" = edge_transition_v3 ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $X $2 -c pepflowww_amd/csrc/$F.hip -o /tmp/expx_$F.o || exit 1
  objs=$(echo $objs | sed "s#pepflowww_amd/lib/$F.o#/tmp/expx_$F.o#")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs
for W in $3; do
timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$1] [$2] $W', round(d['ms_per_step'],4), d.get('kernel_share_of_step'))"
done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
