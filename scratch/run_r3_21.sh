#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_21
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
    acc[(n,r["Grid_Size"],r["Workgroup_Size"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:28]:
    print("%-62s grid %-10s wg %-5s n=%4d avg %8.1f us total %9.1f"%(k[0],k[1],k[2],len(v),sum(v)/len(v),sum(v)))
PY
