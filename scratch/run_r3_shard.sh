#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/shard
timeout 600 python bench.py --force-sharded --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/shard/req.json 2> gpurun_out/shard/req.err; tail -c 600 gpurun_out/shard/req.json; echo
FAIRREC_SHARD_SCHEDULE=item_owner timeout 600 python bench.py --force-sharded --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/shard/own.json 2> gpurun_out/shard/own.err; tail -c 300 gpurun_out/shard/own.json; echo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/shard/tr.json 2> gpurun_out/shard/tr.err; tail -c 300 gpurun_out/shard/tr.json; echo
timeout 600 python bench.py --workload nfcf100m --steps 20 --warmup 3 --no-cpu-baseline --force-sharded > gpurun_out/shard/nfcf.json 2> gpurun_out/shard/nfcf.err; tail -c 400 gpurun_out/shard/nfcf.json; tail -3 gpurun_out/shard/nfcf.err
