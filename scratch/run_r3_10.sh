#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_10
mkdir -p $O
cd $R
L=$R/scratch/lib
for stp in 250 251 260; do
FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so TRACE_STEP=$stp TRACE_OUT=$O/g$stp.npz python scratch/graph_trace.py 2>$O/g$stp.err | tail -8
done
