#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
echo "=== focf tests"; timeout 1500 python -m pytest tests/test_focf_hip.py -q -m gpu -p no:cacheprovider -k "runs or prefetch" 2>&1 | tail -4 | cut -c1-300
A="--item-dist grouped --steps 200 --graph-only"
python scratch/bench_brief.py $A
python scratch/bench_brief.py $A --sweep 0
FAIRREC_RUNS_SWEEP_SPLIT=30 python scratch/bench_brief.py $A
