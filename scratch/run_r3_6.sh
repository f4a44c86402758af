#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --graph-only > $O/trace.log 2>&1
python3 $R/profiles/trace_window.py $O/trace 200 > $O/graph_window.txt 2>&1
cat $O/graph_window.txt | head -60
