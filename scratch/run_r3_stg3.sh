#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stg3
mkdir -p $O
cd $R
L=$R/scratch/lib
timeout 600 python -m pytest tests/test_focf_hip.py -x -q -m gpu -k "in_launch_prepare" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run first$rep -
run last$rep stlast
run mid$rep stmid
FAIRREC_FOCF_STAGED=0 run sorted$rep -
done
export FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so
TRACE_STEP=260 timeout 300 python scratch/graph_trace.py > $O/trace_first.txt 2>$O/trace_first.err; cat $O/trace_first.txt
export FAIRREC_HIP_LIB=$L/libfairrec_hip_tracelast.so
TRACE_STEP=260 timeout 300 python scratch/graph_trace.py > $O/trace_last.txt 2>$O/trace_last.err; cat $O/trace_last.txt
