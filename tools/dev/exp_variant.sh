#!/bin/bash
# build a variant lib with -DPF_EXP_NOLO for edge_transition_v3 only, bench ET
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $1 -c pepflowww_amd/csrc/edge_transition_v3.hip -o /tmp/et3x.o
objs=""; for f in selftest linear edge_transition ipa_attn node_ops flow_step encode node_track train_fwd backward ipa_bwd full_atom; do objs="$objs pepflowww_amd/lib/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/et3x.o
for w in cfg4; do timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(\"$w\", round(d[\"ms_per_step\"],4), d[\"roofline\"][\"avg_launch_us\"])"; done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
