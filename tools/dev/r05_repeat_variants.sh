#!/bin/bash
# dev: tools/dev/r05_ipa_repeat.py over several library builds:  r05_repeat_variants.sh OUTTAG name1 name2 ...  ("main" = the in-tree lib)
mkdir -p gpurun_out
tag=$1; shift
out=gpurun_out/${tag}.txt; : > $out
for name in "$@"; do
  echo "=== $name" >> $out
  if [ "$name" = main ]; then timeout 300 python tools/dev/r05_ipa_repeat.py 2>&1 | grep -v amdgpu.ids >> $out
  else PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$name.so timeout 300 python tools/dev/r05_ipa_repeat.py 2>&1 | grep -v amdgpu.ids >> $out; fi
  echo "$name: $(tail -1 $out)"
done
