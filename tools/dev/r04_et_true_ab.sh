# dev: the TRUE A/B of the fragment-ordered pair tensor: the EdgeTransition objects of the commit before it (libpf_etold.so, PF_ET_ZFRAG=0)
# against the tree's library with its default (z_frag on) and with PF_ET_ZFRAG=0 (the new build's own fallback path)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for P in fp32 f16; do for C in old new0 new1 old new0 new1; do
  unset PF_LIB_PATH PF_ET_ZFRAG
  if [ $C = old ]; then export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_etold.so PF_ET_ZFRAG=0; fi
  if [ $C = new0 ]; then export PF_ET_ZFRAG=0; fi
  timeout 300 python bench.py --workload cfg4 --precision $P $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 $P $C', round(d['ms_per_step'],4), 'ET', round(d['roofline']['avg_launch_us'],1))"
done; done
