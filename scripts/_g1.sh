cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 11000 7800; do
timeout -k 10 300 python3 scripts/run_steps.py --h $h --fused 1 --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130
for p in 200 224 240 256 288; do
NXS_DEBUG_PATCHES=1 NXS_EXP_T320=1 timeout -k 10 300 python3 scripts/run_steps.py --h $h --fused 1 --steps 100 --patch-nodes $p 2>&1 | grep -v amdgpu.ids | grep "patches: P=\|ms/step" | cut -c1-150
done; done
NXS_EXP_T320=1 timeout -k 10 300 python3 scripts/run_steps.py --h 11000 --fused 1 --steps 3 --patch-nodes 240 --compare-fused 0 2>&1 | grep -v amdgpu.ids | tail -1
