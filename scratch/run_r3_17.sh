#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager', d['ms_per_step']*1e3, d['roofline']['kernel_us'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph', d['ms_per_step']*1e3, d['config']['launch_modes_timed'], d['roofline']['kernel_us'])"
done
