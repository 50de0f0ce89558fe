#!/bin/bash
# dev-only what-if builds of the persistent EdgeTransition kernel (timing only; results may be WRONG)
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
for EXP in "" $EXPS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${EXP//,/ } -c pepflowww_amd/csrc/edge_transition_v3.hip -o /tmp/et3_exp.o 2>/dev/null || exit 1
  objs=$(ls pepflowww_amd/lib/*.o | grep -v edge_transition_v3)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/et3_exp.o
  for W in ${WL:-cfg4}; do
  python bench.py --workload $W --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('EXP[$EXP] $W', round(d['ms_per_step'],3), 'ms/step  ET us', round(d['roofline']['avg_launch_us'],1))"
  done
done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
