#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stg5
mkdir -p $O
cd $R
L=$R/scratch/lib
timeout 900 python -m pytest tests/test_focf_hip.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2 3; do
run intask$rep -
run entry$rep entry
run mid$rep stmid
run w5_$rep w5
FAIRREC_FOCF_STAGED=0 run sorted$rep -
done
