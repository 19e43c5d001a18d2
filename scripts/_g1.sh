cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2b_suite6.log 2>&1; echo "suite rc $?"
tail -3 gpurun_out/r2b_suite6.log
