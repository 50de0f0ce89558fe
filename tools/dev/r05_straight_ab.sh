#!/bin/bash
# dev: the STRAIGHT-LINE prologue (role tests folded away at compile time: `const int role = 0`, no helper waves) with the query points handed
# over through LDS (source of commit e0f1086) and in registers (in-tree source): bitwise repeat test, N launches, both kernel forms.
# Variants: sed the two lines in a copy of ipa_split.hip, tools/dev/build_variant_src.sh straight_lds / straight_regs (not kept in the tree).
N=${N:-20000}
for round in 1 2; do
  for lib in straight_lds straight_regs; do
    lp=$PWD/pepflowww_amd/lib/variants/libpf_$lib.so
    r0=$(PF_LIB_PATH=$lp PF_REPEAT_LAUNCHES=$N python tools/dev/r05_repeat_old_form.py 2>&1 | tail -1)
    r1=$(PF_LIB_PATH=$lp PF_REPEAT_LAUNCHES=$N python -m pytest tests/test_gpu_fresh_process.py -q -k "many_launches and True" 2>&1 | grep -E "launches differ|passed" | tail -1)
    echo "round $round, library $lib, $N launches: form k_from_s=0: $r0 | form k_from_s=1: $r1"
  done
done
