cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --h 15600 --resident 2>&1 | tail -8
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so timeout -k 10 120 python3 scripts/phase_timing.py --h 15600 --resident 2>&1 | tail -8
