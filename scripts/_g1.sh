cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 11000 7800; do for i in 1 2; do
NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_old.so timeout -k 10 300 python3 scripts/run_steps.py --h $h --fused 1 --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
timeout -k 10 300 python3 scripts/run_steps.py --h $h --fused 1 --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
done; done
