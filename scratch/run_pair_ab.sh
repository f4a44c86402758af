#!/bin/bash
# A/B of the pair gather in the pipelined item-run step (same box): tests first, then grouped bench with both libraries
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_focf_hip.py -x -q -m gpu -k "pipelined or runs" 2>&1 | tail -5 > gpurun_out/r4/pair_tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-shapes --item-dist grouped --graph-only 2>/dev/null | tail -1 > gpurun_out/r4/pair_on_$i.json
  FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_nopair.so python bench.py --no-cpu-baseline --no-shapes --item-dist grouped --graph-only 2>/dev/null | tail -1 > gpurun_out/r4/pair_off_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/pair_o*_?.json')):
    try:
        d=json.load(open(f)); print(f, d.get('ms_per_step'), d.get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r4/pair_tests.log
