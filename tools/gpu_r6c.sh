#!/bin/bash
# Round 6, big grids: A/B of library builds (LIBS="a b": gpurun_lib_<name>.so, tools/mk_big.sh) x chunks per thread of the LEAN step kernels
# (CPTS="1 2 4" -> ARCLE_BIG_CPT): us per step of the C3 mix and of the light / heavy op families, instruction counts per wave (PMC, one
# kernel name per plane size), then the big-grid GPU tests on the first library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIBS=${LIBS:-cpt}; CPTS=${CPTS:-"1 2"}; TAG=${TAG:-r6c}
for lib in $LIBS; do for cpt in $CPTS; do
  export ARCLE_HIP_LIB=$R/gpurun_lib_$lib.so ARCLE_BIG_CPT=$cpt
  echo "== lib $lib ARCLE_BIG_CPT=$cpt"
  timeout 600 python tools/bigbench.py --sizes 40x40,64x64,127x127 --envs 16384 2>&1 | grep envs | cut -c1-150
  for ops in 0-9 20-23 24-27; do timeout 300 python tools/bigbench.py --sizes 40x40,64x64 --envs 16384 --ops $ops 2>&1 | grep envs | cut -c1-80; done
done; done | tee $O/${TAG}_time.txt
echo "== PMC per wave"
for lib in $LIBS; do for cpt in $CPTS; do for ops in 0-9 20-23 0-34; do for size in 40x40 64x64; do
  export ARCLE_HIP_LIB=$R/gpurun_lib_$lib.so ARCLE_BIG_CPT=$cpt
  echo "-- lib $lib ARCLE_BIG_CPT=$cpt ops $ops $size"
  bash tools/gpu_kpmc.sh python $R/tools/bigbench.py --eager --sizes $size --envs 4096 --steps 12 --ops $ops 2>&1 | grep -v "reset\|^kernel"
done; done; done; done | tee $O/${TAG}_pmc.txt
export ARCLE_HIP_LIB=$R/gpurun_lib_${LIBS%% *}.so; unset ARCLE_BIG_CPT
timeout 900 python -m pytest tests/test_big_hip.py -q -m gpu --tb=short -x 2>&1 | tail -5
