#!/bin/bash
# the driver's command is `python3 bench.py --gpus 1 --steps 20 --warmup 5`: what do the launch modes give on a 0.6 ms timed region?
run() { echo "== $*: $(env $1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-workloads --graph-only ${@:2} 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('launch'))")"; }
for rep in 1 2; do
run X=1 --launch library
run X=1 --launch graph
run ROC_ACTIVE_WAIT_TIMEOUT=1000000 --launch library
run ROC_ACTIVE_WAIT_TIMEOUT=1000000 --launch graph
done
