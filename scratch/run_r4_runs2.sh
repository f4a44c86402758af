#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== runs kernel: new test"; timeout 900 python -m pytest tests/test_focf_hip.py -q -m gpu -p no:cacheprovider -k "runs_step" 2>&1 | tail -8 | cut -c1-300
A="--item-dist grouped --steps 200 --graph-only"
python scratch/bench_brief.py $A
python scratch/bench_brief.py $A --sweep 0
FAIRREC_FOCF_RUNS=0 python scratch/bench_brief.py $A
FAIRREC_FOCF_RUNS=0 python scratch/bench_brief.py $A --sweep 0
for v in rs2 rw16 rw4; do
  FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_$v.so python scratch/bench_brief.py $A
  FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_$v.so python scratch/bench_brief.py $A --sweep 0
done
python scratch/bench_brief.py --steps 200 --graph-only
