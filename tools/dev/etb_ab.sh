#!/bin/bash
# dev-only: the EdgeTransition backward chain kernel (csrc/et_bwd.hip) rebuilt with other tile sizes / register budgets ON the GPU box
# and timed in the training bench: tools/dev/etb_ab.sh "64 2" "32 2" "32 3" ...   (pairs per workgroup, workgroups per CU)
B="python bench.py --workload cfg5 --no-cpu-baseline --no-secondary --no-modes --no-per-call --steps 30 --warmup 5"
for cfg in "$@"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPF_ETB_P=$1 -DPF_ETB_WGS=$2 -c pepflowww_amd/csrc/et_bwd.hip -o pepflowww_amd/lib/et_bwd.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so pepflowww_amd/lib/*.o
  for i in 1 2; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P=$1 WGS=$2 ms_per_step', round(d['ms_per_step'],3))"; done
done
