#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_16
mkdir -p $O
cd $R
timeout 900 python bench.py --workload pfcn10m --steps 20 --warmup 5 > $O/pfcn10m.json 2> $O/pfcn10m.err; tail -c 1500 $O/pfcn10m.json; tail -3 $O/pfcn10m.err
timeout 900 python bench.py --workload nfcf100m --nfcf-users 1000001 --nfcf-items 100001 --steps 20 --warmup 5 > $O/nfcf_small.json 2> $O/nfcf_small.err; tail -c 800 $O/nfcf_small.json; tail -3 $O/nfcf_small.err
timeout 1500 python bench.py --workload fairgo10m --users 1000001 --items 100001 --steps 5 --warmup 3 > $O/fairgo_small.json 2> $O/fairgo_small.err; tail -c 1500 $O/fairgo_small.json; tail -3 $O/fairgo_small.err
