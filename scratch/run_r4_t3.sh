#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== focf + fairgo + trainer tests"; timeout 2400 python -m pytest tests/test_focf_hip.py tests/test_fairgo_hip.py tests/test_trainer_hip.py tests/test_e2e_hip.py -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | cut -c1-300
A="--item-dist grouped --steps 200 --graph-only"
python scratch/bench_brief.py $A
python scratch/bench_brief.py $A --sweep 0
FAIRREC_FOCF_RUNS=0 python scratch/bench_brief.py $A
python scratch/bench_brief.py --steps 200 --graph-only
