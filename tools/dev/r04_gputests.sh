cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${1:-r04}_gputest.log
cat gpurun_out/${1:-r04}_gputest.log
