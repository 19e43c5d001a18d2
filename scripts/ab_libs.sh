# A/B of library builds on the GPU box: bash scripts/ab_libs.sh <tag> <steps> lib1.so lib2.so ...   (each: scripts/run_steps.py --mesh 2km, twice)
set -o pipefail
TAG=$1; STEPS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/${TAG}.log
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (pass $rep)" >> gpurun_out/${TAG}.log
  NXS_DYN_LIBRARY=$GRAFT_REPO_ROOT/nextsim_amd/csrc/$lib timeout -k 10 300 python3 scripts/run_steps.py --mesh 2km --steps $STEPS ${AB_ARGS} >> gpurun_out/${TAG}.log 2>&1 || exit 1
done
done
grep -E "^==|ms/step" gpurun_out/${TAG}.log | sed -E "s/shape_mem.*triangles, //; s/element-updates.*prep_ms.: ([0-9.]+).*substeps_ms.: ([0-9.]+).*ring_flush_ms.: ([0-9.]+).*/prep \1 substeps \2 flush \3/" | cut -c1-200
