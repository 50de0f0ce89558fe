# dev: same-box A/B of the fragment-ordered att_vt planes (f16 mode, the separate-projection path): libpf_head3.so (before) vs the tree's library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call --precision f16"
for F in head3 new head3 new; do
  if [ $F = new ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$F.so; fi
  timeout 300 python bench.py --workload cfg3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 f16 lib=$F', round(d['ms_per_step'],4))"
  PF_FUSED_PROJ=0 timeout 300 python bench.py --workload cfg4 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 f16 (projection launch) lib=$F', round(d['ms_per_step'],4))"
done
