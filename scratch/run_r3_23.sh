#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_23
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --workload pfcn10m --steps 6 --warmup 4 > $O/trace.log 2>&1
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob("$O/trace/**/*kernel_trace.csv",recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f))]
acc=collections.defaultdict(list)
for r in rows[-3000:]:
    n=r["Kernel_Name"].split("(")[0].replace("void fr::","")[:60]
    acc[(n,r["Grid_Size_X"],r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print("%-50s grid %-9s y %-2s n=%4d avg %8.1f us"%(k[0],k[1],k[2],len(v),sum(v)/len(v)))
PY
grep -h "^{" $O/trace.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['filter_pass_ms'], d['config']['dis_pass_ms'])"
