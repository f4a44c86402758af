#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pre
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3 4 5; do
run pre_$rep -
run nopre_$rep nopre
run head_$rep head
done
for rep in 1 2; do
run zipf_pre_$rep - --item-dist zipf
run zipf_nopre_$rep nopre --item-dist zipf
run zipf_head_$rep head --item-dist zipf
done
