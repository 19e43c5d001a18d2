cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NXS_DEBUG_PATCHES=1
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --h 11000 2>&1 | grep -v amdgpu | tail -9
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --h 7800 2>&1 | grep -v amdgpu | tail -9
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --mesh 2km 2>&1 | grep -v amdgpu | tail -9
