#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/uniq
mkdir -p $O
cd $R
run() { local tag=$1; shift
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2; do
run uniform_staged_$rep
run unique_staged_$rep --item-dist unique
FAIRREC_FOCF_STAGED=0 run uniform_sorted_$rep
FAIRREC_FOCF_STAGED=0 run unique_sorted_$rep --item-dist unique
done
export FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_trace.so
FAIRREC_FOCF_STAGED=1 TRACE_STEP=260 timeout 300 python scratch/graph_trace.py --item-dist unique > $O/trace_unique_staged.txt 2>$O/err.txt; echo "== unique staged"; cat $O/trace_unique_staged.txt
FAIRREC_FOCF_STAGED=0 TRACE_STEP=260 timeout 300 python scratch/graph_trace.py --item-dist unique > $O/trace_unique_sorted.txt 2>$O/err.txt; echo "== unique sorted"; cat $O/trace_unique_sorted.txt
