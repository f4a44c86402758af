#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for rep in 1 2; do
TAG=eager python scratch/step_bench.py --no-graph 2>/dev/null
MAIN_PRIO=-1 TAG=eager_main_hi python scratch/step_bench.py --no-graph 2>/dev/null
SIDE_PRIO=1 TAG=eager_side_lo python scratch/step_bench.py --no-graph 2>&1 | tail -1
MAIN_PRIO=-1 TAG=graph_main_hi python scratch/step_bench.py 2>/dev/null
done
