#!/bin/bash
# HBM traffic of the big kernels (run ON the GPU box through gpurun): separate rocprofv3 --pmc passes for FETCH_SIZE and
# WRITE_SIZE (only with --kernel-trace) -> gpurun_out/pmc_traffic_<tag>.json (copy to profiles/rNN/pmc_traffic.json)
TAG=${1:-r02}; COMMIT=${2:-unknown}; cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for W in cfg2 cfg4; do
  for P in fp32 f16; do
    [ $W = cfg2 ] && [ $P = f16 ] && continue
    for C in FETCH_SIZE WRITE_SIZE; do
      OUT=gpurun_out/pt_${W}_${P}_$C; rm -rf $OUT
      PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-secondary --no-modes --no-per-call --workload $W --precision $P > $OUT.log 2>&1
    done
  done
done
python - "$TAG" "$COMMIT" <<'PY'
import csv, glob, json, re, sys, collections
tag, commit = sys.argv[1], sys.argv[2]
sys.path.insert(0, ".")
import bench
res = {"_commit": commit, "_src_sha": bench.kernel_src_sha(),
       "_how": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE, separate pass) --kernel-trace --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-graph --workload W --precision P; per-kernel average of Counter_Value (KiB). Correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 128-B requests as 64 B for wide coalesced 16 B/lane reads -> doubled; WRITE_SIZE as reported."}
KEYS = ("edge_transition_v5h_kernel", "edge_transition_v5_kernel", "edge_transition_v4_kernel", "edge_transition_v3_kernel", "edge_transition_kernel", "ipa_attn_kernel", "ipa_scores_kernel", "ipa_scores16_kernel", "ipa_pair_kernel", "ipa_pair_dz_kernel", "ipa_pair_dz16_kernel", "linear_split_kernel", "linear_rows_kernel", "node_tfmr_kernel", "node_head_kernel", "node_head32_kernel")
for W in ("cfg2", "cfg4"):
    for P in ("fp32", "f16"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for C in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(f"gpurun_out/pt_{W}_{P}_{C}/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    for key in KEYS:
                        if key + "<" in k or key + "(" in k or k.strip().endswith(key):
                            if row["Counter_Name"] == C:
                                acc[key][C].append(float(row["Counter_Value"]))
        if not acc:
            continue
        name = W if P == "fp32" else f"{W}_{P}"
        res[name] = {}
        for key, d in acc.items():
            f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
            w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
            res[name][key] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_corrected": int((2 * f + w) * 1024), "launches": len(d["FETCH_SIZE"])}
        # every kernel of OURS in the run, per step: bytes per launch x launches per step (bench.py sums them: whole_step_traffic).
        # The run holds W + K sampler steps + min(K, 10) instrumented plan runs = 1 + 4 + 4 = 9 passes over the plan.
        allk = collections.defaultdict(lambda: collections.defaultdict(list))
        for C in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(f"gpurun_out/pt_{W}_{P}_{C}/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    if "anonymous namespace" in k and "at::" not in k and "rocprim" not in k and row["Counter_Name"] == C and "edge_features" not in k and "node_features" not in k:
                        m_ = re.search(r"(\w+_kernel)\b", k)                  # (a kernel whose ARGUMENT type lives in a namespace: the name is not the last "::" piece)
                        short = m_.group(1) if m_ else k.split("::")[-1].split("(")[0].split("<")[0].strip()
                        allk[short][C].append(float(row["Counter_Value"]))
        per = {}
        for short, d in allk.items():
            n = len(d["FETCH_SIZE"])
            if n < 9:                       # set-up kernels (bind_context, sampler init): not part of a step
                continue
            f = sum(d["FETCH_SIZE"]) / max(1, n)
            w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
            lps = n / 9.0 if short != "sampler_step_kernel" else n / 5.0
            per[short] = {"hbm_bytes_corrected": int((2 * f + w) * 1024), "launches_per_step": int(round(lps))}
        res[name]["_per_step"] = per
        sk = "ipa_scores_kernel" if "ipa_scores_kernel" in res[name] else "ipa_scores16_kernel"
        pk = next((k for k in ("ipa_pair_dz_kernel", "ipa_pair_dz16_kernel") if k in res[name]), "ipa_pair_kernel")
        if sk in res[name] and pk in res[name]:
            a, b = res[name][sk], res[name][pk]
            res[name]["ipa_two_kernel_form"] = {"hbm_bytes_corrected": a["hbm_bytes_corrected"] + b["hbm_bytes_corrected"], "note": sk + " + " + pk + " (one pf_ipa_attn_fwd call)"}
json.dump(res, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pt_cfg*
