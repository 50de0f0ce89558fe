cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
# (no -x: a failure must not hide the rest; the full log is kept so a rare failure can be read afterwards)
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/${1:-r04}_gputest_full.log
tail -15 gpurun_out/${1:-r04}_gputest_full.log > gpurun_out/${1:-r04}_gputest.log
cat gpurun_out/${1:-r04}_gputest.log
