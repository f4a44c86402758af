#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/valu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/full -- python3 $R/profiles/pmc_step.py > $O/full.log 2>&1
export PMC_BATCH=64 PMC_SWEEP=123
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/sweeponly -- python3 $R/profiles/pmc_step.py > $O/sweeponly.log 2>&1
export PMC_BATCH=8192 PMC_SWEEP=0 PMC_STEPS=60
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/nosweep -- python3 $R/profiles/pmc_step.py > $O/nosweep.log 2>&1
cd $R
for d in full sweeponly nosweep; do echo "== $d"; tail -1 $O/$d.log; python profiles/pmc_summary.py $O/$d focf_step 20; done
