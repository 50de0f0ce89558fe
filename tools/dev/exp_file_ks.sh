#!/bin/bash
# dev-only: several flag sets for ONE source, ks4 rows matching a pattern.  exp_file_ks.sh <stem> <grep-pattern> "<flags1>" "<flags2>" ...
R=$GRAFT_REPO_ROOT; cd $R
F=$1; PAT=$2; shift 2
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
X=""; [ $F = edge_transition_v3 ] && X=-fno-slp-vectorize
for FL in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $X $FL -c pepflowww_amd/csrc/$F.hip -o /tmp/expx.o || exit 1
  objs=""; for f in pepflowww_amd/lib/*.o; do [ "$f" != "pepflowww_amd/lib/$F.o" ] && objs="$objs $f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/expx.o
  echo "== [$FL]"; bash tools/dev/ks4.sh ${PREC:-fp32} | grep "$PAT\|ms_per"
done
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
