#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_3
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2; do
  for cap in 0 1024 1536 2048 2560 3072; do FAIRREC_STEP_SWCAP=$cap run swcap${cap}_$rep -; done
done
FAIRREC_STEP_SWCAP=2048 FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so TRACE_OUT=$O/trace_cap2048.npz python scratch/step_trace.py > $O/trace_cap2048.log 2>&1
tail -9 $O/trace_cap2048.log
FAIRREC_STEP_SWCAP=2048 python -m pytest tests/test_focf_hip.py -m gpu -x -q 2>&1 | tail -2
