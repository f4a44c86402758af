#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/uniq2
mkdir -p $O
cd $R
export FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_trace.so
for d in unique uniform; do for m in 1 0; do
FAIRREC_FOCF_STAGED=$m TRACE_STEP=260 TRACE_OUT=$O/tr_${d}_$m.npz timeout 300 python scratch/graph_trace.py --item-dist $d > $O/trace_${d}_$m.txt 2>$O/err.txt
echo "== $d staged=$m"; tail -4 $O/trace_${d}_$m.txt; python scratch/trace_order.py $O/tr_${d}_$m.npz
done; done
