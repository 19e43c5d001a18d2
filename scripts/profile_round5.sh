# Round-5 profile of the hot path on the GPU box (through gpurun, from the repo root):   bash scripts/profile_round5.sh [tag]   -> gpurun_out/<tag>_*
#   1. the bench line as the driver runs it (N = 1)                                  -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (no aux legs)            -> <tag>_prof/stats
#   3. FETCH_SIZE / WRITE_SIZE passes (separate, counters only) over one 2 km step   -> <tag>_prof/pmc
#   4. SQ counters (waves, VALU instructions, busy cycles; separate pass)            -> <tag>_prof/sq
#   5. a rank of two (rank 0's half of the 2 km mesh, mailboxes looped back) on k_substep_pair<HALO>: kernel statistics + the two counter passes -> <tag>_prof/half, pmc_half
#   6. the bench line at N = 2 with both ranks on THIS GPU (protocol rehearsal: its figures bound the overhead, they are no scaling measurement) -> <tag>_bench_n2_one_gpu.json
#   6b. a rank of eight (rank 0's eighth, looped back) in the resident loop with the one-launch smoother: kernel statistics -> <tag>_prof/eighth
#   7. kernel statistics of the resident launches: 182 k (one rank of eight), 367 k (one rank of four)  -> <tag>_prof/stats182, stats367
# A step that times out ends the script (no further GPU step after a killed one).
set -o pipefail
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
run() { echo "== $1" >&2; shift; timeout -k 10 600 "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; return $rc; }
HALF="--mesh 2km --nparts 2 --rank 0 --loopback --opt pair_regs=-1 --opt fused=3 --opt halo_fused=1"
run bench python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux --no-live-pmc > $OUT/bench_under_prof.json 2> $OUT/bench_under_prof.err || exit 1
run fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/FETCH_SIZE -o f -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/fetch.log 2>&1 || exit 1
run write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/WRITE_SIZE -o w -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/write.log 2>&1 || exit 1
run sq rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o q -- python3 scripts/run_steps.py --mesh 2km --steps 1 --graph 0 > $OUT/sq.log 2>&1 || echo "SQ pass failed (counter names?): see sq.log"
run half rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/half -o s -- python3 scripts/run_steps.py $HALF --steps 10 > $OUT/half.log 2>&1 || echo "half-mesh pass failed: see half.log"
run half_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_half/FETCH_SIZE -o f -- python3 scripts/run_steps.py $HALF --steps 1 --graph 0 > $OUT/half_fetch.log 2>&1 || echo "half-mesh FETCH pass failed"
run half_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_half/WRITE_SIZE -o w -- python3 scripts/run_steps.py $HALF --steps 1 --graph 0 > $OUT/half_write.log 2>&1 || echo "half-mesh WRITE pass failed"
NXS_BENCH_SKIP_RCCL=1 run bench_n2 python3 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_n2_one_gpu.json 2> gpurun_out/${TAG}_bench_n2_one_gpu.err || echo "N = 2 rehearsal failed: see the .err file"   # (self-launched: no torchrun)
EIGHTH="--mesh 2km --nparts 8 --rank 0 --loopback --opt fused=4 --opt halo_fused=1 --opt resident_wide=1"
run eighth rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eighth -o s -- python3 scripts/run_steps.py $EIGHTH --steps 10 > $OUT/eighth.log 2>&1 || echo "rank-of-eight pass failed: see eighth.log"
run res182 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats182 -o s -- python3 scripts/run_steps.py --h 15600 --steps 20 --fused 4 > $OUT/res182.log 2>&1 || exit 1
run res367 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats367 -o s -- python3 scripts/run_steps.py --h 11000 --steps 20 --fused 4 > $OUT/res367.log 2>&1 || exit 1
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
ls $OUT $OUT/stats | head -40
