cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > gpurun_out/r2_parity3.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_parity3.log
for rep in 1 2; do
for m in h15600 10km 2km; do
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_old.so python scripts/run_steps.py --mesh $m --steps 60 --torch-first 2>&1 | tail -1 | cut -c1-120 | sed "s/^/old /"
python scripts/run_steps.py --mesh $m --steps 60 --torch-first 2>&1 | tail -1 | cut -c1-120 | sed "s/^/new /"
done; done
