#!/bin/bash
# dev: another build of the library with ONE source compiled with extra flags -> pepflowww_amd/lib/variants/libpf_<name>.so
#   tools/dev/build_variant.sh nop0 ipa_split.hip -DPJ_DMA_NOPS=0        (run with PF_LIB_PATH=.../libpf_nop0.so for same-box A/B)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../pepflowww_amd"
mkdir -p lib/variants
extra=""
case $src in edge_transition_v3.hip) extra="-fno-slp-vectorize";; edge_transition_v4.hip|edge_transition_v5.hip) extra="-fno-slp-vectorize -Wno-inline-asm";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra "$@" -c csrc/$src -o lib/variants/${src%.hip}_$name.o 2>/dev/null
objs=$(ls lib/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libpf_$name.so $objs lib/variants/${src%.hip}_$name.o
rm lib/variants/${src%.hip}_$name.o
echo built lib/variants/libpf_$name.so
