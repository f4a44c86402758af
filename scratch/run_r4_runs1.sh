#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== runs kernel: new test"; timeout 900 python -m pytest tests/test_focf_hip.py -q -m gpu -p no:cacheprovider -x -k "runs_step" 2>&1 | tail -30 | cut -c1-300
echo "=== runs kernel: goldens"; timeout 1500 python -m pytest tests/test_focf_hip.py -q -m gpu -p no:cacheprovider -k "golden and runs" 2>&1 | tail -40 | cut -c1-300
echo "=== bench grouped (runs kernel)"; timeout 600 python bench.py --item-dist grouped --steps 200 --graph-only --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1500
echo "=== bench grouped (chain, FAIRREC_FOCF_RUNS=0)"; FAIRREC_FOCF_RUNS=0 timeout 600 python bench.py --item-dist grouped --steps 200 --graph-only --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1500
bash scratch/run_r4_soak.sh b 40
