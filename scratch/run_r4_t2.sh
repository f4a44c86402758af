#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
echo "=== trainer + focf tests"; timeout 1500 python -m pytest tests/test_trainer_hip.py tests/test_focf_hip.py tests/test_e2e_hip.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25 | cut -c1-300
A="--item-dist grouped --steps 200 --graph-only"
python scratch/bench_brief.py $A
FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_rw8.so python scratch/bench_brief.py $A
FAIRREC_FOCF_RUNS=0 python scratch/bench_brief.py $A
echo "=== bench default (with shapes)"; timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r4/bench_default.json')); print(d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('other_batch_shapes'), indent=1))"
echo "=== fairgo bench (1M x 100k)"; timeout 1200 python bench.py --workload fairgo10m --users 1000001 --items 100001 --steps 3 --warmup 3 > gpurun_out/r4/fairgo1m.json 2> gpurun_out/r4/fairgo1m.err; python -c "
import json; d=json.load(open('gpurun_out/r4/fairgo1m.json')); print(d['ms_per_step'], d['config'])"
tail -3 gpurun_out/r4/fairgo1m.err
