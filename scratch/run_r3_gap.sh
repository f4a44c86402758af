#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gap
mkdir -p $O
cd $R
export FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_trace.so
for st in 255 260; do
TRACE_STEP=$st timeout 300 python scratch/graph_trace.py > $O/trace_$st.txt 2>$O/err.txt; cat $O/trace_$st.txt
done
unset FAIRREC_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --graph-only > $O/kt.log 2>&1
cd $R
python profiles/trace_window.py $O/kt 2>/dev/null | head -8
