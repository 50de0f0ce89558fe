#!/bin/bash
# PMC passes over one kernel of the bench (run ON the GPU box through gpurun):
#   tools/pmc_kernel.sh <kernel-substring> <workload> [extra bench args]
# Counters only with --kernel-trace (never with sys/hip/hsa trace domains).  Summary -> gpurun_out/pmc_<kernel>_<workload>.txt
K=${1:-edge_transition_v5_kernel}; W=${2:-cfg4}; shift 2
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_${PMC_TAG:-$(echo $K | cut -d, -f1)}_${W}
rm -rf $OUT && mkdir -p $OUT
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"
 "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  PF_BENCH_NO_SCLK=1 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python bench.py --workload $W --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary --no-modes --no-per-call "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, kerns = sys.argv[1], sys.argv[2].split(",")          # several kernels (comma separated) share the passes
rows = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
with open(out + ".txt", "w") as fh:
    for kern in kerns:
        acc = collections.defaultdict(list)
        for row in rows:
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        fh.write(f"== {kern}\n"); print("==", kern)
        for k in sorted(acc):
            v = acc[k]
            line = f"{k:36s} mean/launch {sum(v)/len(v):16.1f}   n={len(v)}"
            print(line); fh.write(line + "\n")
PY
find $OUT -name "*kernel_trace*" -delete; find $OUT -name "*.csv" -size +4M -delete
