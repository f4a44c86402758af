#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/flaky3
loop() { local tag=$1 n=$2; shift 2; local f=0
  for k in $(seq 1 $n); do
    "$@" > gpurun_out/flaky3/$tag$k.log 2>&1
    if grep -qE "[0-9]+ failed" gpurun_out/flaky3/$tag$k.log; then f=$((f+1)); echo -n F; else echo -n .; rm -f gpurun_out/flaky3/$tag$k.log; fi
  done; echo " $tag failures: $f / $n"; }
loop pre4 8 python -m pytest tests/test_abi.py tests/test_bench_contract_hip.py tests/test_bench_launch.py tests/test_dropout_hip.py tests/test_eval_hip.py tests/test_fairgo_hip.py -q -m gpu -p no:cacheprovider
loop bench 8 python -m pytest tests/test_bench_contract_hip.py tests/test_fairgo_hip.py -q -m gpu -p no:cacheprovider
loop evalf 8 python -m pytest tests/test_dropout_hip.py tests/test_eval_hip.py tests/test_fairgo_hip.py -q -m gpu -p no:cacheprovider
