#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/wts
mkdir -p $O
cd $R
L=$R/scratch/lib
show() { python - "$1" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], d["config"].get("filter_pass_ms"), d["config"].get("dis_pass_ms"),
      {n:(round(v["ms"]*1e3,1), v["launches"]) for n,v in r["kernels_filter_pass"].items() if n.startswith("bn")},
      {n:(round(v["ms"]*1e3,1), v["launches"]) for n,v in r["kernels_dis_pass"].items() if n.startswith("bn")})
PY
}
for rep in 1 2; do
unset FAIRREC_HIP_LIB
timeout 600 python bench.py --workload pfcn10m --no-cpu-baseline > $O/plain_$rep.json 2> $O/plain_$rep.err; show $O/plain_$rep.json
export FAIRREC_HIP_LIB=$L/libfairrec_hip_wts.so
timeout 600 python bench.py --workload pfcn10m --no-cpu-baseline > $O/wts_$rep.json 2> $O/wts_$rep.err; show $O/wts_$rep.json
done
unset FAIRREC_HIP_LIB
timeout 600 python -m pytest tests/test_focf_hip.py -x -q -m gpu > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -1
