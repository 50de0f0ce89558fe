#!/bin/bash
# dev-only: rebuild linear.hip with extra flags into the library, run tools/dev/proj_bench.py, restore
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $1 -c pepflowww_amd/csrc/linear.hip -o /tmp/linx.o
objs=""; for f in selftest edge_transition edge_transition_v3 ipa_attn node_ops flow_step encode node_track train_fwd backward ipa_bwd full_atom; do objs="$objs pepflowww_amd/lib/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/linx.o
echo "flags [$1]: $(python tools/dev/proj_bench.py ${2:-1024} 2>/dev/null | tail -1)"
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
