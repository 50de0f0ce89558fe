#!/bin/bash
# dev-only (run ON the GPU box): EdgeTransition compile-flag variants, bench timings of the kernel.  exp_et.sh "<flags1>" "<flags2>" ...
R=$GRAFT_REPO_ROOT; cd $R
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
for FL in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $FL -c pepflowww_amd/csrc/edge_transition_v3.hip -o /tmp/expx.o || exit 1
  objs=""; for f in pepflowww_amd/lib/*.o; do [ "$f" != "pepflowww_amd/lib/edge_transition_v3.o" ] && objs="$objs $f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/expx.o
  for P in ${PRECS:-fp32 f16}; do
    python bench.py --no-cpu-baseline --no-secondary --precision $P 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$FL] $P', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1))"
  done
done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
