cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for H in 1 0; do for P in 1 0; do
echo "== helpers=$H fused_proj=$P"; PF_PROJ_HELPERS=$H PF_FUSED_PROJ=$P timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_graph_equals_eager or full_size_shard" 2>&1 | tail -4
done; done
