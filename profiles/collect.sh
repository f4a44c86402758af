#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r01'
# 1) --kernel-trace --stats of the default `bench.py` command (same command whose JSON line is reported)
# 2) --pmc FETCH_SIZE and 3) --pmc WRITE_SIZE (separate passes, TCC slot limits) of profiles/pmc_workload.py,
#    which first runs table_flush_kernel on a known byte count to calibrate the counters in our access pattern.
# 4) --kernel-trace of `bench.py --graph-only`, reduced by profiles/trace_window.py to the hipGraph replay window
# Raw output lands in gpurun_out/<round>/ (scratch); `python profiles/summarize.py <round>` then writes the
# tracked summaries into profiles/.
set -u
ROUND=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_write.log 2>&1
grep -h '^{' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line.json
# 4) kernel trace of the hipGraph replay alone -> per-kernel durations and a two-step timeline inside the replay window
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --graph-only > $OUT/trace.log 2>&1
python3 $R/profiles/trace_window.py $OUT/trace 200 > $OUT/graph_window.txt 2>&1
ls -R $OUT | head -30
