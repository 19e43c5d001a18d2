cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 scripts/run_steps.py --h 15600 --steps 50 --fused 4 --compare-fused 1 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do
timeout -k 10 200 python3 scripts/run_steps.py --h 15600 --steps 200 --fused 4 2>&1 | grep -v amdgpu.ids | tail -1
done
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --h 15600 --resident 2>&1 | tail -7
