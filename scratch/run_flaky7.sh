#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/flaky7
f=0
for k in $(seq 1 40); do
  FAIRREC_TEST_NO_GRAPH=${NOGRAPH:-} python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "(full_batch and data_parallel) or trainer" > gpurun_out/flaky7/r$k.log 2>&1
  if grep -qE "[0-9]+ failed|xfailed" gpurun_out/flaky7/r$k.log; then f=$((f+1)); echo -n F; grep -E "^XFAIL" gpurun_out/flaky7/r$k.log | cut -c1-700; else echo -n .; rm -f gpurun_out/flaky7/r$k.log; fi
done; echo " failures: $f / 40"
