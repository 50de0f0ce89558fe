#!/bin/bash
# dev: same-box A/B of library variants in the step:  tools/dev/ab_lib.sh <tag> <workload> <precision> <name1> <name2> ...  ("main" = the in-tree library)
# prints ms_per_step and the HIP-event time of every C-ABI call of the plan (kernel_us) for each library, two rounds (A B A B)
tag=$1; wl=$2; prec=$3; shift 3
mkdir -p gpurun_out; out=gpurun_out/${tag}.txt; : > $out
for round in 1 2; do
  for name in "$@"; do
    if [ "$name" = main ]; then lp=""; else lp=$PWD/pepflowww_amd/lib/variants/libpf_$name.so; fi
    PF_LIB_PATH=$lp PF_BENCH_NO_SCLK=1 python bench.py --workload $wl --precision $prec --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-per-call 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$name" "$round" >> $out <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
ks = {"pf_edge_transition_fwd": "ET", "pf_ipa_attn_fwd": "attn", "pf_node_tfmr_fwd": "tfmr", "pf_node_head_fwd": "head", "pf_linear_fwd": "lin"}
share = d.get("kernel_share_of_step", {})
print(f"{sys.argv[1]:12s} round {sys.argv[2]}: {d['ms_per_step']:.4f} ms/step   ET launch {d['roofline']['avg_launch_us']:.1f} us  attn launch {d['roofline_other']['avg_launch_us']:.1f} us   shares {share}")
PY
  done
done
cat $out
