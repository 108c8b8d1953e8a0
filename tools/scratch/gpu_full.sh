cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r3/gputests_r3a.log 2>&1; rc=$?
tail -5 gpurun_out/r3/gputests_r3a.log
exit $rc
