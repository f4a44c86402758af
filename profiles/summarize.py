#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<round>/) into the tracked summaries:
  profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py`
  profiles/<round>_pmc.md             FETCH_SIZE / WRITE_SIZE per kernel, calibration and corrected bytes
  profiles/pmc_traffic.json           {kernel: corrected HBM-side bytes per launch}  (read by bench.py)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", rnd)
# (on the GPU box collect.sh passes a directory under gpurun_out/ as the destination -- the raw traces are too large to travel
# back -- and the files are copied into profiles/ here afterwards: `python profiles/summarize.py r06 --take`)
dst = os.path.join(ROOT, "profiles")
if len(sys.argv) > 2 and sys.argv[2] == "--take":
    n = 0
    for f in sorted(glob.glob(os.path.join(src, "summary", "*"))):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        n += 1
    print(f"{n} files from {src}/summary into profiles/")
    sys.exit(0)
if len(sys.argv) > 2:
    dst = sys.argv[2]
    os.makedirs(dst, exist_ok=True)


def one(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)   # gpurun merges runs into the same directory: take the newest


shutil.copy(one("stats/**/*_kernel_stats.csv"), os.path.join(dst, f"{rnd}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_line.json")):
    shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{rnd}_bench_under_rocprof.json"))
if os.path.exists(os.path.join(src, "graph_window.txt")):
    shutil.copy(os.path.join(src, "graph_window.txt"), os.path.join(dst, f"{rnd}_graph_window.txt"))


def counters(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"]
            k = k.split("(")[0].replace("void ", "").split("<")[0].replace("fr::", "")
            agg[k].append(float(r["Counter_Value"]))
    return agg


fetch = counters(one("pmc_fetch/**/*_counter_collection.csv"), "FETCH_SIZE")
write = counters(one("pmc_write/**/*_counter_collection.csv"), "WRITE_SIZE")
N, D = 1_000_000, 64
known = N * (3 * D * 4 + 4)               # bytes table_flush_kernel reads AND writes in the calibration launches
f_cal = known / (1024 * sum(fetch["table_flush_kernel"]) / len(fetch["table_flush_kernel"]))
w_cal = known / (1024 * sum(write["table_flush_kernel"]) / len(write["table_flush_kernel"]))
lines = [f"# PMC traffic, round {rnd}", "",
         "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, `profiles/pmc_workload.py`.",
         f"Calibration on table_flush_kernel (known {known} B read and written per launch, dword-per-lane 256-B rows):",
         f"FETCH_SIZE*1024 under-reports by x{f_cal:.3f} (guide: exactly 1/2 on gfx950 wide reads), "
         f"WRITE_SIZE*1024 by x{w_cal:.3f}.", "",
         "Corrected bytes per launch = counter * 1024 * calibration factor (mean of the last 100 launches).", "",
         "| kernel | launches | FETCH_SIZE (KB) | WRITE_SIZE (KB) | read MB | written MB | total MB |", "|---|---|---|---|---|---|---|"]
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    f = fetch.get(k, [0.0])[-100:]
    w = write.get(k, [0.0])[-100:]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    rb, wb = fm * 1024 * f_cal, wm * 1024 * w_cal
    traffic[k] = int(rb + wb)
    lines.append(f"| {k} | {len(fetch.get(k, []))} | {fm:.1f} | {wm:.1f} | {rb / 1e6:.2f} | {wb / 1e6:.2f} | {(rb + wb) / 1e6:.2f} |")
open(os.path.join(dst, f"{rnd}_pmc.md"), "w").write("\n".join(lines) + "\n")
# what the counters were collected ON: bench.py compares this with the sources it runs and flags a stale figure
import hashlib
_src = os.path.join(ROOT, "recbole-fairrec_amd", "csrc")
traffic["_kernel_source_sha1"] = hashlib.sha1(b"".join(open(os.path.join(_src, f), "rb").read() for f in
                                                       ("focf_step.hip", "focf_ws.hpp", "common.hpp"))).hexdigest()
traffic["_round"] = rnd
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))

# ---- round 3: plain bench lines, the other workloads, MFMA counters, the wave timeline -------------------------------------
for name in ("bench.json", "bench_steps20.json", "bench_zipf.json", "bench_grouped_pipe.json", "bench_grouped_runs.json", "bench_grouped_chain.json", "bench_grouped_fused.json",
             "pfcn10m.json", "nfcf100m.json", "nfcf1m.json", "fairgo10m.json", "pmc_mfma.md", "wave_trace.txt", "eval_probe.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        if name.endswith(".json"):      # keep the JSON line only
            lines_ = [l for l in open(f).read().splitlines() if l.startswith("{")]
            if lines_:
                open(os.path.join(dst, f"{rnd}_{name}"), "w").write(lines_[-1] + "\n")
        else:
            shutil.copy(f, os.path.join(dst, f"{rnd}_{name}"))
if os.path.exists(os.path.join(src, "pmc_mfma_pfcn.json")):
    shutil.copy(os.path.join(src, "pmc_mfma_pfcn.json"), os.path.join(dst, "pmc_mfma_pfcn.json"))
for tag in ("pfcn", "nfcf", "fairgo"):
    files = glob.glob(os.path.join(src, f"stats_{tag}/**/*_kernel_stats.csv"), recursive=True)
    if files:
        shutil.copy(max(files, key=os.path.getmtime), os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
