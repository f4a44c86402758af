#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r03'
# 1) --kernel-trace --stats of the default `bench.py` command (same command whose JSON line is reported)
# 2) --pmc FETCH_SIZE and 3) --pmc WRITE_SIZE (separate passes, TCC slot limits) of profiles/pmc_workload.py,
#    which first runs table_flush_kernel on a known byte count to calibrate the counters in our access pattern.
# 4) --kernel-trace of `bench.py --graph-only`, reduced by profiles/trace_window.py to the hipGraph replay window
# 5) plain (un-profiled) bench lines: default, --steps 20, --item-dist zipf / grouped (chain and one-launch step)
# 6) the other BASELINE configs: bench.py --workload pfcn10m / nfcf100m / fairgo10m, each plain and under --kernel-trace --stats
# 7) --pmc SQ_VALU_MFMA_BUSY_CYCLES ... of profiles/pmc_pfcn.py (MFMA utilisation of the PFCN step)
# Raw output lands in gpurun_out/<round>/ (scratch); `python profiles/summarize.py <round>` then writes the
# tracked summaries into profiles/.
set -u
ROUND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-workloads > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/profiles/pmc_workload.py > $OUT/pmc_write.log 2>&1
grep -h '^{' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --graph-only --no-workloads > $OUT/trace.log 2>&1
python3 $R/profiles/trace_window.py $OUT/trace 200 > $OUT/graph_window.txt 2>&1
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
# (round 6: the headline is the library's step loop; the item-complete variants below keep rounds 4-5's hipGraph replay so that
# their numbers stay comparable, the library-loop figure of each shape is in bench.json's other_batch_shapes)
python3 bench.py --no-cpu-baseline --graph-only --item-dist zipf > $OUT/bench_zipf.json 2>/dev/null
python3 bench.py --no-cpu-baseline --graph-only --launch graph --item-dist grouped > $OUT/bench_grouped_pipe.json 2>/dev/null
FAIRREC_FOCF_PIPE=0 python3 bench.py --no-cpu-baseline --graph-only --launch graph --item-dist grouped > $OUT/bench_grouped_runs.json 2>/dev/null
FAIRREC_FOCF_RUNS=0 python3 bench.py --no-cpu-baseline --graph-only --launch graph --item-dist grouped > $OUT/bench_grouped_chain.json 2>/dev/null
python3 bench.py --no-cpu-baseline --graph-only --launch graph --item-dist grouped --force-fused > $OUT/bench_grouped_fused.json 2>/dev/null
python3 bench.py --workload pfcn10m --steps 20 --warmup 5 > $OUT/pfcn10m.json 2> $OUT/pfcn10m.err
python3 bench.py --workload nfcf100m --steps 20 --warmup 5 > $OUT/nfcf100m.json 2> $OUT/nfcf100m.err
python3 bench.py --workload nfcf100m --nfcf-users 1000001 --nfcf-items 100001 --steps 20 --warmup 5 > $OUT/nfcf1m.json 2> $OUT/nfcf1m.err
timeout 2400 python3 bench.py --workload fairgo10m --steps 3 --warmup 3 > $OUT/fairgo10m.json 2> $OUT/fairgo10m.err
# 8) round 5: the plugin surface (Trainer._train_epoch through the loaders), the device shuffle, the VALU issue rates
STEPS=1024 python3 scratch/trainer_bench.py device > $OUT/trainer_bench_device.txt 2>&1
python3 scratch/randperm_bench.py > $OUT/randperm_bench.txt 2>&1
hipcc -O3 --offload-arch=gfx950 scratch/valu_rates.hip -o /tmp/valu_rates > /dev/null 2>&1 && /tmp/valu_rates > $OUT/valu_rates.txt 2>&1
hipcc -O3 --offload-arch=gfx950 scratch/replay_bench.hip -o /tmp/replay_bench > /dev/null 2>&1 && /tmp/replay_bench 128 3000 > $OUT/replay_bench.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pfcn -- python3 $R/bench.py --workload pfcn10m --steps 10 --warmup 5 > $OUT/pfcn_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nfcf -- python3 $R/bench.py --workload nfcf100m --nfcf-users 1000001 --nfcf-items 100001 --steps 10 --warmup 5 > $OUT/nfcf_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fairgo -- python3 $R/bench.py --workload fairgo10m --users 1000001 --items 100001 --steps 3 --warmup 3 > $OUT/fairgo_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -- python3 $R/profiles/pmc_pfcn.py > $OUT/pmc_mfma.log 2>&1
python3 $R/profiles/pmc_mfma_summary.py $OUT/pmc_mfma $OUT/pmc_mfma_pfcn.json > $OUT/pmc_mfma.md 2>&1
# (the trace build is made from the same sources here, so that its C ABI is the product library's)
make -C $R/recbole-fairrec_amd/csrc VARIANT=trace EXTRA=-DFR_STEP_TRACE=1 -j8 > $OUT/trace_build.log 2>&1
if [ -f $R/scratch/lib/libfairrec_hip_trace.so ]; then
  cd $R && FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_trace.so python3 scratch/step_trace.py > $OUT/wave_trace.txt 2>&1
fi
# 9) round 6: where an evaluation batch's time goes
python3 scratch/eval_probe.py 500000 > $OUT/eval_probe.txt 2>&1
# 10) the tracked summaries are made HERE (gpurun brings back at most 64 MiB; the raw traces are several times that)
cd $R && python3 profiles/summarize.py $ROUND $OUT/summary > $OUT/summarize.log 2>&1
{ echo "# rocprofv3 --kernel-trace of bench.py --workload nfcf100m at 1 M x 100 k, reduced by profiles/step_window.py (anchor table_lookup_kernel)"; python3 profiles/step_window.py $OUT/stats_nfcf table_lookup_kernel 8; } > $OUT/summary/${ROUND}_nfcf_step.txt 2>&1
{ echo "# rocprofv3 --kernel-trace of bench.py --workload pfcn10m, reduced by profiles/step_window.py (anchor bpr_outer_kernel): the FILTER pass of a step (the densest window is the aging phase, which runs filter passes only)"; python3 profiles/step_window.py $OUT/stats_pfcn bpr_outer_kernel 8; } > $OUT/summary/${ROUND}_pfcn_step.txt 2>&1
for f in trainer_bench_device.txt randperm_bench.txt valu_rates.txt replay_bench.txt; do [ -s $OUT/$f ] && cp $OUT/$f $OUT/summary/${ROUND}_$f; done
du -sh $OUT/* | sort -h | tail -12
rm -rf $OUT/stats $OUT/stats_pfcn $OUT/stats_nfcf $OUT/stats_fairgo $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/trace
for f in $OUT/*.err; do [ -s $f ] && { echo "== $f"; grep -v amdgpu.ids $f | tail -3; }; done
tail -5 $OUT/summarize.log
ls $OUT/summary
