# what the GPU box gives a run: CPUs, cgroup quota, memory, NUMA (for the CPU baseline's placement notes)
echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo "no cgroup v2 cpu.max"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null
lscpu | egrep "Model name|Socket|Core|Thread|NUMA|MHz" | head -14
cat /proc/self/status | egrep "Cpus_allowed_list|Mems_allowed_list"
cat /proc/pressure/cpu 2>/dev/null | head -3
