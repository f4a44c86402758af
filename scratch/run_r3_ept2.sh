#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ept2
mkdir -p $O
cd $R
L=$R/scratch/lib
timeout 900 python -m pytest tests/test_focf_hip.py -x -q -m gpu > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -2
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run p1_$rep -
run p2_$rep p2
run ept1_$rep ept1
run zipf_p1_$rep - --item-dist zipf
run zipf_p2_$rep p2 --item-dist zipf
run zipf_ept1_$rep ept1 --item-dist zipf
done
