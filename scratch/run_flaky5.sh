#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/flaky5
loop() { local tag=$1 n=$2; shift 2; local f=0
  for k in $(seq 1 $n); do
    "$@" > gpurun_out/flaky5/$tag$k.log 2>&1
    if grep -qE "[0-9]+ failed|xfailed" gpurun_out/flaky5/$tag$k.log; then f=$((f+1)); echo -n F; else echo -n .; rm -f gpurun_out/flaky5/$tag$k.log; fi
  done; echo " $tag failures: $f / $n"; }
K='(full_batch and data_parallel) or trainer'
FAIRREC_TEST_PG_SYNC=0 loop nosync_a 12 python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "$K"
FAIRREC_TEST_PG_SYNC=1 loop sync_a 12 python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "$K"
FAIRREC_TEST_PG_SYNC=0 loop nosync_b 12 python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "$K"
FAIRREC_TEST_PG_SYNC=1 loop sync_b 12 python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "$K"
