# scratch wrapper for one gpurun call (rocprofv3 wants /tmp as working directory while it starts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2b_suite7.log 2>&1; echo "suite rc $?"
tail -3 gpurun_out/r2b_suite7.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
