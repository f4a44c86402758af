#!/bin/bash
# A/B of the chained launches of fr_focf_steps_many (FAIRREC_FOCF_CHAIN) on the product library and on A/B builds
# usage: scratch/chain_ab.sh <outdir> [lib ...]   (lib = a path under scratch/lib, or "main")
out=$1; shift
mkdir -p $out
for lib in "$@"; do
  for ch in 0 1; do
    tag=$(basename $lib .so)_ch$ch
    if [ "$lib" = main ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$PWD/$lib; fi
    FAIRREC_FOCF_CHAIN=$ch timeout 300 python bench.py --no-shapes --no-cpu-baseline $BENCH_ARGS > $out/$tag.json 2> $out/$tag.err
    echo "$tag rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1])
    print(d['ms_per_step'], d['config']['launch_modes_timed'], d['config']['final_loss'])
except Exception as e: print('no json', e)
")"
  done
done
