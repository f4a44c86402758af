#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
A="--item-dist grouped --steps 200 --graph-only"
for v in mb16 w16; do
FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_$v.so python scratch/bench_brief.py $A
FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_$v.so python scratch/bench_brief.py $A --sweep 0
done
