#!/bin/bash
# dev-only (run ON the GPU box): rebuild ONE source with extra flags, print per-kernel rocprof averages at cfg4 (ks4.sh), restore.
#   tools/dev/exp_ks4.sh <source-stem> "<flags>" [fp32|f16]
R=$GRAFT_REPO_ROOT; cd $R
F=$1; FLAGS=$2; P=${3:-fp32}
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
X=""; [ $F = edge_transition_v3 ] && X=-fno-slp-vectorize
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $X $FLAGS -c pepflowww_amd/csrc/$F.hip -o /tmp/expx.o || exit 1
objs=""; for f in pepflowww_amd/lib/*.o; do [ "$f" != "pepflowww_amd/lib/$F.o" ] && objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/expx.o
echo "== $F [$FLAGS] $P"
W=${W:-cfg4} bash tools/dev/ks4.sh $P
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
