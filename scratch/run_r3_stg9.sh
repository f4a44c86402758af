#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stg9
mkdir -p $O
cd $R
L=$R/scratch/lib
timeout 900 python -m pytest tests/test_focf_hip.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run first$rep -
run mid$rep stmid
FAIRREC_FOCF_STAGED=0 run sorted$rep -
done
run zipf_first - --item-dist zipf
run zipf_mid stmid --item-dist zipf
FAIRREC_FOCF_STAGED=0 run zipf_sorted - --item-dist zipf
export FAIRREC_HIP_LIB=$L/libfairrec_hip_trace.so
for st in 255 260; do
FAIRREC_FOCF_STAGED=1 TRACE_STEP=$st timeout 300 python scratch/graph_trace.py > $O/trace_staged_$st.txt 2>$O/err.txt; echo "== staged $st"; cat $O/trace_staged_$st.txt
FAIRREC_FOCF_STAGED=0 TRACE_STEP=$st timeout 300 python scratch/graph_trace.py > $O/trace_sorted_$st.txt 2>$O/err.txt; echo "== sorted $st"; cat $O/trace_sorted_$st.txt
done
