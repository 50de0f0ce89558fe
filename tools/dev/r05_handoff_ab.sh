#!/bin/bash
# dev: same-box A/B of the query-point hand-off: the bitwise repeat test (B x L = 64 x 128, N launches) on the library variant that hands the
# query points through wave-private LDS (the form of rounds 4 - 5, pepflowww_amd/lib/variants/libpf_lds_handoff.so) and on the in-tree
# library (registers + cross-lane reads), both kernel forms (k_from_s = 0 / 1), two rounds.  The variant is not kept in the tree; rebuild it with
#   git show e0f1086:pepflowww_amd/csrc/ipa_split.hip > /tmp/x.hip && bash tools/dev/build_variant_src.sh lds_handoff ipa_split.hip /tmp/x.hip
N=${N:-20000}
for round in 1 2; do
  for lib in lds_handoff main; do
    if [ "$lib" = main ]; then lp=""; else lp=$PWD/pepflowww_amd/lib/variants/libpf_$lib.so; fi
    r0=$(PF_LIB_PATH=$lp PF_REPEAT_LAUNCHES=$N python tools/dev/r05_repeat_old_form.py 2>&1 | tail -1)
    r1=$(PF_LIB_PATH=$lp PF_REPEAT_LAUNCHES=$N python -m pytest tests/test_gpu_fresh_process.py -q -k "many_launches" 2>&1 | grep -E "launches differ|passed" | tail -1)
    echo "round $round, library $lib, $N launches: form k_from_s=0: $r0 | form k_from_s=1: $r1"
  done
done
