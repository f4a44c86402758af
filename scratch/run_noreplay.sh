#!/bin/bash
mkdir -p gpurun_out/r4
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-shapes --item-dist grouped --graph-only 2>/dev/null | tail -1 > gpurun_out/r4/nr_base_$i.json
  FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_noreplay.so python bench.py --no-cpu-baseline --no-shapes --item-dist grouped --graph-only 2>/dev/null | tail -1 > gpurun_out/r4/nr_diag_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/nr_*_?.json')):
    try:
        d=json.load(open(f)); print(f, d.get('ms_per_step'), d.get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
