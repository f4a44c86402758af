#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/waves
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run w6_$rep -
run w7_$rep w7
run w8_$rep w8
run w5_$rep w5
done
run zipf_w6 - --item-dist zipf
run zipf_w7 w7 --item-dist zipf
