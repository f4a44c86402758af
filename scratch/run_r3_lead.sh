#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lead
mkdir -p $O
cd $R
run() { local tag=$1; shift
  TAG=$tag timeout 300 python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt; }
for rep in 1 2; do
run lead100_$rep
FAIRREC_STEP_LEAD=90 run lead90_$rep
FAIRREC_STEP_LEAD=80 run lead80_$rep
FAIRREC_STEP_LEAD=60 run lead60_$rep
FAIRREC_STEP_LEAD=30 run lead30_$rep
done
run fresh_nosweep --age -1 --sweep 0
run fresh_sweep --age -1
run unique --item-dist unique
