#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_13
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
python -m pytest tests/test_focf_hip.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  run new_$rep -
  run base_$rep base
done
run new_zipf - --item-dist zipf
run new_s20 - --steps 20 --warmup 5
FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so TRACE_OUT=$O/trace.npz python scratch/step_trace.py > $O/trace.log 2>&1
grep -E "kernel span|^sweeper|^interaction|SIMDs seen|phases|clock" $O/trace.log
