#!/bin/bash
# the runtime's graph switches on the driver's command (20-step hipGraph replay)
run() { echo "== $1: $(env $1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --graph-only --no-shapes 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('launch'))")"; }
for rep in 1 2; do
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=1
done
