#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
bash scratch/run_r4_soak.sh a 40
echo "=== frontier"; timeout 900 python -m pytest tests/test_fairgo_hip.py -q -m gpu -p no:cacheprovider -k frontier 2>&1 | tail -25 | cut -c1-400
echo "=== full gpu suite"; timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -30 | cut -c1-400
