"""MFMA utilisation per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE and
SQ_INSTS_VALU_MFMA_MOPS_F32 (profiles/collect.sh).  util = MFMA-busy cycles summed over the chip's 1024 SIMDs / (1024 x the
launch's cycles), the launch's cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs; MI355X_MICROARCH.md).
Writes a markdown table to stdout and profiles/pmc_mfma_pfcn.json (read by bench.py --workload pfcn10m)."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void fr::", "").split("<")[0]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key)
        cnt[name] += 1
rows, out = [], {}
tot_busy = tot_cyc = 0.0
for name, c in acc.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    tot_busy += busy
    tot_cyc += cyc
    if busy > 0:
        util = busy / (1024.0 * cyc) if cyc else 0.0
        rows.append((name, cnt[name], cyc / cnt[name], util, c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / cnt[name]))
        out[name] = round(util, 4)
rows.sort(key=lambda r: -r[2] * r[1])
print("| kernel | launches | cycles per launch | MFMA busy / (1024 SIMDs x cycles) | MFMA MOPS F32 per launch |")
print("|---|---|---|---|---|")
for r in rows:
    print("| %s | %d | %.0f | %.3f | %.3g |" % r)
whole = tot_busy / (1024.0 * tot_cyc) if tot_cyc else 0.0
print("\nwhole run (every kernel of the filter + discriminator passes): MFMA busy %.4f of the chip's SIMD cycles" % whole)
out["whole_step"] = round(whole, 4)
out["source"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -- python3 profiles/pmc_pfcn.py"
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
