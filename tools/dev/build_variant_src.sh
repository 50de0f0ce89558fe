#!/bin/bash
# dev: like build_variant.sh, but the variant replaces <file.hip> by ANOTHER source file:  build_variant_src.sh <name> <replaced.hip> <source path> [flags]
set -e
name=$1; rep=$2; src=$3; shift 3
cd "$(dirname "$0")/../../pepflowww_amd"
mkdir -p lib/variants
cp "$src" csrc/zz_variant_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c csrc/zz_variant_$name.hip -o lib/variants/v_$name.o 2>/dev/null
rm csrc/zz_variant_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libpf_$name.so $(ls lib/*.o | grep -v "/${rep%.hip}.o") lib/variants/v_$name.o
rm lib/variants/v_$name.o
echo built lib/variants/libpf_$name.so
