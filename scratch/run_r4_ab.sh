#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== fill kernel (default)"; timeout 600 python scratch/graph_idem.py fwd_bwd full pieces 2>&1 | grep "^stage\|^bad\|Error\|error" | cut -c1-300
echo "=== FAIRREC_SCATTER_MEMSET=1"; FAIRREC_SCATTER_MEMSET=1 timeout 600 python scratch/graph_idem.py fwd_bwd full pieces 2>&1 | grep "^stage\|^bad\|Error\|error" | cut -c1-300
echo "=== memset node"; timeout 300 python scratch/memset_node.py 2>&1 | grep -v Warn | tail -20
echo "=== hunt with poison (fill kernel)"; HUNT_N=6 HUNT_PG=0 HUNT_POISON=1 timeout 300 python scratch/nan_hunt.py 2>&1 | grep "^iter\|^done" | cut -c1-400
echo "=== e2e tests"; timeout 1500 python -m pytest tests/test_e2e_hip.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -40 | cut -c1-400
echo "=== nfcf new tests"; timeout 600 python -m pytest tests/test_nfcf_hip.py -q -m gpu -p no:cacheprovider -k "frozen_table or no_slot or global_df" 2>&1 | tail -15 | cut -c1-300
