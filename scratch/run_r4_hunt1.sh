#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
( HUNT_N=6 HUNT_PG=0 HUNT_POISON=1 timeout 300 python scratch/nan_hunt.py ) > gpurun_out/r4/hunt1_a.log 2>&1
( HUNT_N=6 HUNT_PG=1 HUNT_POISON=1 FAIRREC_RCCL_QUIESCE_S=0.3 timeout 300 python scratch/nan_hunt.py ) > gpurun_out/r4/hunt1_b.log 2>&1
( HUNT_N=25 HUNT_PG=1 HUNT_BIG=1 timeout 600 python scratch/nan_hunt.py ) > gpurun_out/r4/hunt1_c.log 2>&1
tail -n 12 gpurun_out/r4/hunt1_*.log
