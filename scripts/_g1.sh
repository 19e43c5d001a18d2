cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r2b_bench4.json 2> gpurun_out/r2b_bench4.err; echo "bench rc $?"
