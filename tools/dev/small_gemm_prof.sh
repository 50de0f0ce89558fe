#!/bin/bash
# dev-only: true per-shape durations of the row-sized products (run ON the GPU box)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
rm -rf gpurun_out/sg
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sg -- python tools/dev/small_gemm_bench.py > gpurun_out/sg.log 2>&1
f=$(find gpurun_out/sg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:50],r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# consecutive runs of the same (kernel, grid) = one shape of the bench
out=[]; prev=None
for s,e,n,gx,gy,gz in rows:
    k=(n,gx,gy,gz)
    if 'gemm_f32' not in n: continue
    if k!=prev: out.append([k,[]]); prev=k
    out[-1][1].append(e-s)
for k,v in out:
    if len(v)>=5: print(f"{sum(v)/len(v)/1e3:8.1f} us x{len(v):3d}  min {min(v)/1e3:6.1f}  grid {k[1]}x{k[2]}x{k[3]}  {k[0]}")
PY
rm -rf gpurun_out/sg
