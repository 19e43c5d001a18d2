# scratch wrapper for one gpurun call (rocprofv3 wants /tmp as working directory while it starts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NXS_DEBUG_PATCHES=1 timeout -k 10 600 python3 scripts/check_partitions.py 2km 8 2>&1 | grep -v amdgpu.ids | grep "possible"
