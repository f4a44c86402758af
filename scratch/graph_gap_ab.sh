#!/bin/bash
# Where do the ~8 us between the last kernel of one replay of a captured step and the first kernel of the next go?  The NFCF
# step (12 nodes) under the runtime's graph switches.
run() { echo "== $*"; env "$@" python3 bench.py --workload nfcf100m --nfcf-users 1000001 --nfcf-items 100001 --steps 200 --warmup 10 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms_per_step', d['ms_per_step'])"; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run HIP_FORCE_DEV_KERNARG=0
run FAIRREC_GRAPH_MAILBOX=0
# Result (round 6, 1 x MI355X, 200 steps): 0.1467 default | 0.1449 packet capture off | 0.1466 on | 0.1470 graph queues | 0.1544 skip
# release scope | 0.1456 kernarg copy opt off | 0.1452 HDP flush WA off | 0.1450 dev kernarg off | 0.1467 with the batch refresh as
# an ordinary launch in front of the replay instead of the graph's first node (scratch/graph_mailbox_r6.patch).  The ~8 us the
# kernel trace shows between two replays are the profiler's; un-profiled the step is within 1 us of the sum of its kernels.
