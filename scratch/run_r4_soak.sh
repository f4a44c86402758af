#!/bin/bash
# the sequence that reproduced the round-3 flake (1-2 of 14-40 processes on one box in three): the 1-rank RCCL full-batch test,
# then the trainer test, in ONE pytest process; N fresh processes.  usage: run_r4_soak.sh <tag> <N>
cd ${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}; n=${2:-40}
mkdir -p gpurun_out/r4/soak_$tag
f=0
for k in $(seq 1 $n); do
  python -m pytest tests/test_fairgo_hip.py -q -m gpu -rx -p no:cacheprovider -k "(full_batch and data_parallel) or trainer" > gpurun_out/r4/soak_$tag/r$k.log 2>&1
  if grep -qE "2 passed" gpurun_out/r4/soak_$tag/r$k.log && ! grep -qE "failed|xfailed|error" gpurun_out/r4/soak_$tag/r$k.log; then echo -n .; rm -f gpurun_out/r4/soak_$tag/r$k.log; else f=$((f+1)); echo -n F; fi
done; echo " soak $tag: failures $f / $n on $(hostname) $(date -u +%H:%M)"
HUNT_N=30 HUNT_PG=1 HUNT_BIG=1 HUNT_POISON=1 FAIRREC_RCCL_QUIESCE_S=0.3 timeout 600 python scratch/nan_hunt.py 2>&1 | grep "^iter\|^done" | cut -c1-300
