cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in libnxsdyn_old.so libnxsdyn.so; do
echo $lib
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/$lib timeout -k 10 200 python3 scripts/run_steps.py --h 15600 --steps 200 --fused 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-110
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/$lib timeout -k 10 200 python3 scripts/run_steps.py --mesh 10km --steps 200 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-110
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/$lib timeout -k 10 200 python3 scripts/run_steps.py --h 11000 --steps 100 --fused 1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c54-110
done; done
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_flip_set.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
