#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 600 python scratch/graph_idem.py > gpurun_out/r4/idem1.log 2>&1
grep -v "Warning\|warn" gpurun_out/r4/idem1.log | tail -40
