cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env | grep -i rocp
time (timeout -k 10 900 python3 bench.py > gpurun_out/r2b_bench3.json 2> gpurun_out/r2b_bench3.err); echo "bench rc $?"
