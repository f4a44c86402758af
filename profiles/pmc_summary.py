"""Average the PMC counters of a rocprofv3 --pmc pass per kernel (last `tail` dispatches of each kernel).
usage: python profiles/pmc_summary.py <dir> [name-substring] [tail]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 100
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for row in csv.DictReader(open(f)):
    name = row["Kernel_Name"]
    if sub not in name:
        continue
    short = name.split("(")[0][-60:]
    per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    per[short]["dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    meta[short] = (row["Grid_Size"], row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"])
for k, c in per.items():
    print(k, "grid/vgpr/sgpr/lds", meta[k])
    for name, vals in sorted(c.items()):
        v = vals[-tail:]
        print(f"   {name:24s} n={len(vals):5d} avg(last {len(v)}) = {sum(v) / len(v):14.1f}")
