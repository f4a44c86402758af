#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; python -c "
import json; d=json.loads(open('$O/bench20.json').read().strip().splitlines()[-1]); print('steps20', d['ms_per_step']*1e3)"
