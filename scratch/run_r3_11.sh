#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_11
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
python -m pytest tests/test_focf_hip.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  run g16_lw4_$rep -
  FAIRREC_FOCF_LOW_WATER=8 run g16_lw8_$rep -
  FAIRREC_FOCF_GROUP=8 FAIRREC_FOCF_LOW_WATER=4 run g8_lw4_$rep -
  FAIRREC_FOCF_GROUP=24 FAIRREC_FOCF_LOW_WATER=8 run g24_lw8_$rep -
  FAIRREC_FOCF_GROUP=32 FAIRREC_FOCF_LOW_WATER=8 run g32_lw8_$rep -
done
run s20 - --steps 20 --warmup 5
FAIRREC_FOCF_LOW_WATER=8 run s20_lw8 - --steps 20 --warmup 5
