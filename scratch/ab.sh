#!/bin/bash
# usage: scratch/ab.sh <variant> ... : bench.py --steps 200 (uniform, graph replay) per library variant ("" = the product library), twice each
for v in "$@"; do
  for i in 1 2; do
    if [ "$v" = "prod" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$PWD/scratch/lib/libfairrec_hip_$v.so; fi
    python bench.py --steps 200 --no-shapes --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_us'])"
  done
done
