#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --graph-only > $O/trace.log 2>&1
python3 $R/scratch/timeline.py $O/trace 200 236 | tail -64
