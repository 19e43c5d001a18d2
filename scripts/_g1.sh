cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2b_10km_prof -o s -- python3 scripts/run_steps.py --mesh 10km --steps 50 > /dev/null 2>&1
cat gpurun_out/r2b_10km_prof/s_kernel_stats.csv | cut -c1-200
rm -f gpurun_out/r2b_10km_prof/s_kernel_trace.csv
