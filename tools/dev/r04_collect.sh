# one pass over what profiles/r04 holds for a commit: GPU suite, kernel statistics, PMC traffic, bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r04}; COMMIT=${2:-unknown}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${TAG}_gputest.log
bash tools/collect_profiles.sh $TAG $COMMIT > gpurun_out/${TAG}_collect.log 2>&1
tail -3 gpurun_out/${TAG}_gputest.log; ls gpurun_out | grep $TAG
