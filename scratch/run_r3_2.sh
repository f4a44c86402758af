#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_2
mkdir -p $O
cd $R
L=$R/scratch/lib
run() { # tag lib env... -- args
  local tag=$1 lib=$2; shift 2
  if [ "$lib" = "-" ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$L/libfairrec_hip_$lib.so; fi
  TAG=$tag python scratch/step_bench.py "$@" 2>$O/$tag.err | tee -a $O/summary.txt
}
for rep in 1 2; do
  run new_$rep - 
  run nolpt_$rep nolpt
  run wpb1_$rep wpb1
  run wpb2_$rep wpb2
  run prio_$rep prio
  run wpb1prio_$rep wpb1prio
done
OBJ=none run new_objnone -
run new_unique - --item-dist unique
run nolpt_zipf nolpt --item-dist zipf
run new_zipf - --item-dist zipf
run wpb1_zipf wpb1 --item-dist zipf
for lead in 50 80; do FAIRREC_STEP_LEAD=$lead run wpb1_lead$lead wpb1; done
for sw in 62 92; do run wpb1_sweep$sw wpb1 --sweep $sw; done
unset FAIRREC_HIP_LIB
python -m pytest tests/test_focf_hip.py -m gpu -x -q 2>&1 | tail -2
FAIRREC_HIP_LIB=$L/libfairrec_hip_wpb1.so python -m pytest tests/test_focf_hip.py -m gpu -x -q 2>&1 | tail -2
