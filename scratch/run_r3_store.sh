#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/store
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run plain_$rep -
run wt_$rep wt
run wtw_$rep wtw
done
for rep in 1 2; do
run zipf_plain_$rep - --item-dist zipf
run zipf_wt_$rep wt --item-dist zipf
run zipf_wtw_$rep wtw --item-dist zipf
run s20_plain_$rep - --steps 20 --warmup 5
run s20_wt_$rep wt --steps 20 --warmup 5
done
