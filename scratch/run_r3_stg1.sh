#!/bin/bash
# staged (in-launch prepare) FOCF step: parity tests, then benches against the sort-prepared path
cd /root/repo
mkdir -p gpurun_out/stg1
timeout 1200 python -m pytest tests/test_focf_hip.py -x -q -m gpu > gpurun_out/stg1/pytest.log 2>&1
tail -15 gpurun_out/stg1/pytest.log
for mode in 1 0; do
  for rep in 1 2; do
    FAIRREC_FOCF_STAGED=$mode timeout 300 python bench.py --steps 200 --warmup 20 > gpurun_out/stg1/bench_staged${mode}_$rep.json 2> gpurun_out/stg1/bench_staged${mode}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/stg1/bench_staged${mode}_$rep.json").read().strip().splitlines()[-1])
    print("staged=$mode", d["ms_per_step"]*1e3, "us/step", d.get("roofline",{}).get("kernel_us"))
except Exception as e:
    print("staged=$mode failed", e)
    print(open("gpurun_out/stg1/bench_staged${mode}_$rep.err").read()[-2000:])
PY
  done
done
FAIRREC_FOCF_STAGED=1 timeout 300 python bench.py --steps 200 --warmup 20 --item-dist zipf > gpurun_out/stg1/bench_zipf1.json 2> gpurun_out/stg1/bench_zipf1.err
FAIRREC_FOCF_STAGED=0 timeout 300 python bench.py --steps 200 --warmup 20 --item-dist zipf > gpurun_out/stg1/bench_zipf0.json 2> gpurun_out/stg1/bench_zipf0.err
python - <<PY
import json
for m in (1,0):
    try:
        d=json.loads(open(f"gpurun_out/stg1/bench_zipf{m}.json").read().strip().splitlines()[-1])
        print("zipf staged",m, d["ms_per_step"]*1e3)
    except Exception as e:
        print("zipf failed", m, e, open(f"gpurun_out/stg1/bench_zipf{m}.err").read()[-1500:])
PY
