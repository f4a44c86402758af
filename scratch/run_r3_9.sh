#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_9
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2; do
run plain_$rep -
run nt_$rep nt
run sc1_$rep sc1
done
FAIRREC_HIP_LIB=$L/libfairrec_hip_tracesc1.so TRACE_OUT=$O/trace_sc1.npz python scratch/step_trace.py > $O/trace_sc1.log 2>&1
grep -E "kernel span|^sweeper|^interaction|SIMDs seen|phases" $O/trace_sc1.log
