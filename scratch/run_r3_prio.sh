#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prio
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run base_$rep -
run prio_$rep prio
run prioe4_$rep prioe4
run zipf_base_$rep - --item-dist zipf
run zipf_prio_$rep prio --item-dist zipf
done
export FAIRREC_HIP_LIB=$L/libfairrec_hip_traceprio.so
FAIRREC_FOCF_STAGED=1 TRACE_STEP=260 timeout 300 python scratch/graph_trace.py > $O/trace.txt 2>$O/err.txt; cat $O/trace.txt
