#!/bin/bash
# dev-only what-if builds of the EdgeTransition kernel (results are WRONG on purpose; timing only)
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
for EXP in "" "-DPF_EXP_NOGATHER" "-DPF_EXP_NOWSTREAM" "-DPF_EXP_NOGATHER -DPF_EXP_NOWSTREAM" $EXTRA_EXPS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXP -c pepflowww_amd/csrc/edge_transition.hip -o /tmp/et_exp.o || exit 1
  objs=$(ls pepflowww_amd/lib/*.o | grep -v edge_transition)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/et_exp.o
  for t in ${TILES:-641 642 32}; do
    PF_ET_TILE=$t python bench.py --workload ${W:-cfg4} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('EXP[$EXP] TILE=$t', round(d['ms_per_step'],3), 'ms/step  ET us', round(d['roofline']['avg_launch_us'],1))"
  done
done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
