# Round-3 profile of the hot path on the GPU box (through gpurun, from the repo root):   bash scripts/profile_round3.sh [tag]   -> gpurun_out/<tag>_*
#   1. the bench line as the driver runs it (N = 1)                                  -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (no aux legs)            -> <tag>_prof/stats
#   3. FETCH_SIZE / WRITE_SIZE passes (separate, counters only) over one 2 km step   -> <tag>_prof/pmc
#   4. SQ counters (waves, VALU instructions, busy cycles; separate pass)            -> <tag>_prof/sq
#   5. kernel statistics of the resident launches: 182 k (one rank of eight), 367 k (one rank of four)  -> <tag>_prof/stats182, stats367
#   6. kernel statistics of a several-rank step with the exchange inside the kernels: two ranks of one process on this GPU (mailboxes by pointer),
#      one launch per sub-step (k_substep_fused<.., HALO>) and the resident launch (k_substep_resident<.., HALO>)  -> <tag>_prof/halo, halo_resident
# A step that times out ends the script (no further GPU step after a killed one).
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
run() { echo "== $1" >&2; shift; timeout -k 10 600 "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; return $rc; }
run bench python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux --no-live-pmc > $OUT/bench_under_prof.json 2> $OUT/bench_under_prof.err || exit 1
run fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/FETCH_SIZE -o f -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/fetch.log 2>&1 || exit 1
run write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/WRITE_SIZE -o w -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/write.log 2>&1 || exit 1
run sq rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o q -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/sq.log 2>&1 || echo "SQ pass failed (counter names?): see sq.log"
run res182 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats182 -o s -- python3 scripts/run_steps.py --h 15600 --steps 20 --fused 4 > $OUT/res182.log 2>&1 || exit 1
run res367 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats367 -o s -- python3 scripts/run_steps.py --h 11000 --steps 20 --fused 4 > $OUT/res367.log 2>&1 || exit 1
export RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 NXS_RANKS_PER_PROC=2
mkdir -p $OUT/halo_out $OUT/halo_res_out
run halo rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/halo -o s -- python3 tests/mr_worker.py $OUT/halo_out h16000 6 ipc '{"options": {"patch_nodes": 180}}' > $OUT/halo.log 2>&1 || echo "halo pass failed: see halo.log"
run halo_resident rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/halo_resident -o s -- python3 tests/mr_worker.py $OUT/halo_res_out h16000 6 ipc '{"options": {"fused": 4, "patch_nodes": 180}}' > $OUT/halo_res.log 2>&1 || echo "resident halo pass failed: see halo_res.log"
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
ls $OUT $OUT/stats | head -40
