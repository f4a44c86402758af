#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
A="--item-dist grouped --steps 200 --graph-only --sweep 0"
python scratch/bench_brief.py $A
for d in 1 2 3 4; do FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_rd$d.so python scratch/bench_brief.py $A; done
