#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== graph idempotence tests"; timeout 900 python -m pytest tests/test_graph_hip.py -q -m gpu -p no:cacheprovider -k "idempotent" 2>&1 | tail -25 | cut -c1-400
echo "=== e2e tests"; timeout 1500 python -m pytest tests/test_e2e_hip.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 | cut -c1-400
echo "=== fairgo"; timeout 1500 python -m pytest tests/test_fairgo_hip.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25 | cut -c1-400
