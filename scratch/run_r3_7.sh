#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_7
mkdir -p $O
cd $R
L=$R/scratch/lib
for pipe in 24 0; do
FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so TRACE_PIPE=$pipe TRACE_OUT=$O/trace_pipe$pipe.npz python scratch/step_trace.py > $O/trace_pipe$pipe.log 2>&1
echo "== pipe $pipe"; grep -E "kernel span|^sweeper|^interaction|SIMDs seen|phases" $O/trace_pipe$pipe.log
done
