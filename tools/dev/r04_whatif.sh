# dev: what-if builds of the fp32 score kernel (wrong results: the fp32 MFMAs of K Q^T / P V replaced by the 12 / 33-per-32-keys f16
# MFMAs a split-precision form would issue, operands bit-cast) -- what is the most such a form could gain?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-modes --no-per-call"
for W in cfg4 cfg2; do for F in base QK PV QKPV base QK PV QKPV; do
  if [ $F = base ]; then unset PF_LIB_PATH; else export PF_LIB_PATH=$PWD/pepflowww_amd/lib/variants/libpf_wi_$F.so; fi
  timeout 300 python bench.py --workload $W $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W lib=$F', round(d['ms_per_step'],4))"
done; done
