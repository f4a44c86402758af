#!/bin/bash
# round 3, GPU call 1: new parity tests, A/B of the step kernel variants, wave trace, SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_1
mkdir -p $O
cd $R
python -m pytest tests/test_fairgo_hip.py tests/test_focf_hip.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
  for v in base new nolpt; do
    if [ $v = new ]; then unset FAIRREC_HIP_LIB; else export FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_$v.so; fi
    python bench.py --no-cpu-baseline --graph-only > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_${v}_$rep.json"))
    print("$v $rep", d["ms_per_step"] * 1e3, "us/step  kernel", d["roofline"]["kernel_us"])
except Exception as e:
    print("$v $rep failed", e)
PY
  done
done
unset FAIRREC_HIP_LIB
for d in grouped zipf; do
  python bench.py --no-cpu-baseline --graph-only --item-dist $d > $O/bench_new_$d.json 2> $O/bench_new_$d.err
  FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_base.so python bench.py --no-cpu-baseline --graph-only --item-dist $d > $O/bench_base_$d.json 2> $O/bench_base_$d.err
  python -c "
import json
for v in ('base','new'):
    d = json.load(open('$O/bench_%s_$d.json' % v)); print('$d', v, d['ms_per_step']*1e3, d['roofline']['kernel_us'])
"
done
FAIRREC_HIP_LIB=$R/scratch/lib/libfairrec_hip_trace.so TRACE_OUT=$O/trace_new.npz python scratch/step_trace.py > $O/trace_new.log 2>&1
tail -12 $O/trace_new.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq1 -- python3 $R/profiles/pmc_step.py > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS --output-format csv -d $O/pmc_sq2 -- python3 $R/profiles/pmc_step.py > $O/pmc_sq2.log 2>&1
tail -2 $O/pmc_sq1.log $O/pmc_sq2.log
ls $O
