#!/bin/bash
# Round 6, call H: where the mask ingress of the big-grid kernels spends its time — by op family, and instruction counts per wave
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
LIBS=${LIBS:-tree}
for lib in $LIBS; do
  [ $lib = tree ] && unset ARCLE_HIP_LIB || export ARCLE_HIP_LIB=$R/gpurun_lib_$lib.so
  for ing in mask bits bbox; do for ops in 0-34 0-9 10-19 20-23 24-27 28-33; do
    echo -n "lib $lib : "; timeout 600 python tools/bigbench.py --sizes 40x40,64x64 --envs 16384 --steps 12 --ingress $ing --ops $ops 2>&1 | grep envs | cut -c1-62 | tr '\n' '|'; echo
  done; done
done | tee $O/${TAG:-r6h}_mask_by_op.txt
for lib in $LIBS; do
  [ $lib = tree ] && unset ARCLE_HIP_LIB || export ARCLE_HIP_LIB=$R/gpurun_lib_$lib.so
  for ing in mask bits bbox; do for ops in 0-9 20-23; do
    echo "-- lib $lib $ing ops $ops 40x40"
    bash tools/gpu_kpmc.sh python $R/tools/bigbench.py --eager --sizes 40x40 --envs 4096 --steps 12 --ops $ops --ingress $ing 2>&1 | grep "big_step"
  done; done
done | tee $O/${TAG:-r6h}_mask_pmc.txt
