#!/bin/bash
# HBM traffic of the two big kernels (run ON the GPU box through gpurun): separate rocprofv3 --pmc passes for
# FETCH_SIZE and WRITE_SIZE (only with --kernel-trace) -> gpurun_out/pmc_traffic_<tag>.json
TAG=${1:-v3}; cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for W in cfg2 cfg4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=gpurun_out/pt_${W}_$C; rm -rf $OUT
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-secondary --workload $W > $OUT.log 2>&1
  done
done
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
res = {"_how": "rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE, separate pass) --kernel-trace --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graph [--workload cfg4]; per-kernel average of Counter_Value (KiB). Correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 128-B requests as 64 B for wide coalesced 16 B/lane reads -> doubled; WRITE_SIZE as reported."}
for W in ("cfg2", "cfg4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/pt_{W}_{C}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                for key in ("edge_transition_v3_kernel", "edge_transition_kernel", "ipa_attn_kernel"):
                    if key in k and row["Counter_Name"] == C:
                        acc[key][C].append(float(row["Counter_Value"]))
    res[W] = {}
    for key, d in acc.items():
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        res[W][key] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_corrected": int((2 * f + w) * 1024), "launches": len(d["FETCH_SIZE"])}
json.dump(res, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pt_cfg*
