#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stg7
mkdir -p $O
cd $R
L=$R/scratch/lib
export FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so
for rep in 1 2; do
for st in 255 260 263; do
FAIRREC_FOCF_STAGED=1 TRACE_STEP=$st timeout 300 python scratch/graph_trace.py > $O/trace_staged_$st.txt 2>$O/err.txt; echo "== staged $st"; cat $O/trace_staged_$st.txt
FAIRREC_FOCF_STAGED=0 TRACE_STEP=$st timeout 300 python scratch/graph_trace.py > $O/trace_sorted_$st.txt 2>$O/err.txt; echo "== sorted $st"; cat $O/trace_sorted_$st.txt
done
done
