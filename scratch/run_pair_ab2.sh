#!/bin/bash
mkdir -p gpurun_out/r4
FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_ptrace.so python scratch/pipe_trace.py > gpurun_out/r4/pair_trace.log 2>&1
for f in 0 25 50 75 100; do
  FAIRREC_PIPE_SWEEP_FRONT=$f python bench.py --no-cpu-baseline --no-shapes --item-dist grouped --graph-only 2>/dev/null | tail -1 > gpurun_out/r4/pair_front_$f.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/pair_front_*.json')):
    try:
        d=json.load(open(f)); print(f, d.get('ms_per_step'), d.get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r4/pair_trace.log | tail -12
