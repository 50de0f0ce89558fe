#!/bin/bash
# dev-only: what-if builds of the v4 EdgeTransition kernel as separate libraries (pepflowww_amd/lib/variants/), timed with
#   python tools/dev/et_bench.py v4 fp32 pepflowww_amd/lib/variants/libpf_<name>.so
# usage: tools/dev/et_variants.sh name "-DPF_ET4_WHATIF=1" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../.."
mkdir -p pepflowww_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm $flags \
      -c pepflowww_amd/csrc/edge_transition_v4.hip -o pepflowww_amd/lib/variants/et4_$name.o
  objs=$(ls pepflowww_amd/lib/*.o | grep -v edge_transition_v4.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/variants/libpf_$name.so $objs pepflowww_amd/lib/variants/et4_$name.o
  echo built $name
done
