#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/flaky
loop() { local tag=$1 n=$2; shift 2; local f=0
  for k in $(seq 1 $n); do
    "$@" > gpurun_out/flaky/$tag$k.log 2>&1
    if grep -qE "failed|rror" gpurun_out/flaky/$tag$k.log; then f=$((f+1)); cp gpurun_out/flaky/$tag$k.log gpurun_out/flaky/FAIL_$tag$k.log; echo -n F; else echo -n .; rm -f gpurun_out/flaky/$tag$k.log; fi
  done; echo " $tag failures: $f / $n"; }
loop file 12 python -m pytest tests/test_fairgo_hip.py -x -q -m gpu
FAIRREC_NO_OVERLAP=1 loop file_noov 12 python -m pytest tests/test_fairgo_hip.py -x -q -m gpu
loop pfcn 4 python -m pytest tests/test_pfcn_hip.py -x -q -m gpu
