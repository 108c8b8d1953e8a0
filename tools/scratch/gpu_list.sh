cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/r3/counters.txt
wc -c $GRAFT_REPO_ROOT/gpurun_out/r3/counters.txt
