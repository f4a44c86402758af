#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/flaky2
f=0
for k in $(seq 1 90); do
  python -m pytest tests/test_fairgo_hip.py -x -q -m gpu -k "test_fairgo_trainer_pretrain_then_finetune" > gpurun_out/flaky2/r$k.log 2>&1
  if grep -qE "failed|rror" gpurun_out/flaky2/r$k.log; then f=$((f+1)); echo -n F; else echo -n .; rm -f gpurun_out/flaky2/r$k.log; fi
done; echo " failures: $f / 90"
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
