#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_22
mkdir -p $O
cd $R
python -m pytest tests/test_pfcn_hip.py tests/test_graph_hip.py tests/test_trainer_hip.py -m gpu -x -q > $O/t.log 2>&1; grep -E "passed|failed" $O/t.log | tail -2
python bench.py --workload pfcn10m --steps 20 --warmup 5 > $O/pfcn10m.json 2> $O/pfcn10m.err
python - <<PY
import json
d=json.loads(open("$O/pfcn10m.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["filter_pass_ms"], d["config"]["dis_pass_ms"])
print({k:v for k,v in list(d["roofline"]["kernels_filter_pass"].items())[:8]})
PY
