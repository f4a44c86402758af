#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== runs kernel tests"; timeout 900 python -m pytest tests/test_focf_hip.py -q -m gpu -p no:cacheprovider -k "runs" 2>&1 | tail -5 | cut -c1-300
A="--item-dist grouped --steps 200 --graph-only"
python scratch/bench_brief.py $A
python scratch/bench_brief.py $A
FAIRREC_FOCF_RUNS=0 python scratch/bench_brief.py $A
