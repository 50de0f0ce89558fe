# dev: same-box A/B of the second product's forms in the fp32 score kernel with the projection inside ($@ = variant tags under lib/variants, "new" = the tree's library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for W in cfg4 cfg2; do for F in "$@" "$@"; do
  if [ $F = new ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_$F.so; fi
  timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W lib=$F', round(d['ms_per_step'],4))"
done; done
