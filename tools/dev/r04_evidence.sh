# evidence at HEAD for profiles/r04: PMC traffic (every kernel of a step) + SQ counter passes of the big kernels, both modes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r04}; COMMIT=${2:-unknown}
bash tools/pmc_traffic.sh $TAG $COMMIT > gpurun_out/${TAG}_pmc_traffic.log 2>&1
bash tools/pmc_kernel.sh "edge_transition_v4_kernel,ipa_scores_kernel,ipa_pair_dz_kernel,node_head32_kernel" cfg4 --no-modes --no-per-call > /dev/null 2>&1
mv gpurun_out/pmc_edge_transition_v4_kernel,ipa_scores_kernel,ipa_pair_dz_kernel,node_head32_kernel_cfg4.txt gpurun_out/${TAG}_pmc_fp32_cfg4.txt
bash tools/pmc_kernel.sh "edge_transition_v3_kernel,ipa_scores16_kernel" cfg4 --no-modes --no-per-call --precision f16 > /dev/null 2>&1
mv gpurun_out/pmc_edge_transition_v3_kernel,ipa_scores16_kernel_cfg4.txt gpurun_out/${TAG}_pmc_f16_cfg4.txt
rm -rf gpurun_out/pmc_edge_transition_v4_kernel* gpurun_out/pmc_edge_transition_v3_kernel*
ls -la gpurun_out | grep $TAG
