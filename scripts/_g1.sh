cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ring or several or fused or full_size" > gpurun_out/r2_parity4.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_parity4.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_ring -o s -- python3 scripts/run_steps.py --mesh 2km --steps 6 > gpurun_out/r2_ring.log 2>&1; grep -E "k_move_ring|k_ow_tail|k_pingpong" gpurun_out/r2_ring/s_kernel_stats.csv | cut -c1-120
for m in 10km 2km; do python scripts/run_steps.py --mesh $m --steps 60 --torch-first 2>&1 | tail -1 | cut -c1-120; done
