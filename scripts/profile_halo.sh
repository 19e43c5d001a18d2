# kernel statistics of a several-rank step on ONE GPU: two ranks hosted by one process (mailboxes by pointer), the exchange inside the kernels --
# one launch per sub-step (k_substep_fused<.., HALO>) and the resident launch (k_substep_resident<.., HALO>); then the rehearsal of a rank of eight
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT/halo_out $OUT/halo_res_out
export RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 NXS_RANKS_PER_PROC=2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/halo -o s -- python3 tests/mr_worker.py $OUT/halo_out h16000 6 ipc '{"options": {"patch_nodes": 180}}' > $OUT/halo.log 2>&1 || { echo "halo pass failed"; tail -5 $OUT/halo.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/halo_resident -o s -- python3 tests/mr_worker.py $OUT/halo_res_out h16000 6 ipc '{"options": {"fused": 4, "patch_nodes": 180}}' > $OUT/halo_res.log 2>&1 || { echo "resident halo pass failed"; tail -5 $OUT/halo_res.log; exit 1; }
unset RANK WORLD_SIZE MASTER_ADDR MASTER_PORT NXS_RANKS_PER_PROC
for i in 1 2 3; do timeout -k 10 200 python3 scripts/rehearse_rank_of_eight.py 8 2>&1 | grep "^overlap" | cut -c1-260 || exit 1; done
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
