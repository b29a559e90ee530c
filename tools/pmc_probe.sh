#!/bin/bash
# tools/pmc_probe.sh [workload] -- on the GPU box: where a pass's cycles go.  Three separate rocprofv3 --pmc passes
# (never combined with tracing other than --kernel-trace) over `bench.py --workload W`: wave / issue / wait cycles,
# instruction fetch and I-cache, texture-addresser / L1 / TLB stalls.  Summary -> gpurun_out/$ROUND/pmc_probe_W.txt
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-verify}
OUT=$REPO/gpurun_out/${ROUND:-r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-side"
run() { timeout 300 rocprofv3 --pmc "$@" -d $OUT/probe_$W_$n -o p -- $B > $OUT/probe_$W_$n.log 2>&1; }
n=a; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $OUT/probe_${W}_a -o p -- $B > $OUT/probe_${W}_a.log 2>&1
n=b; timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD -d $OUT/probe_${W}_b -o p -- $B > $OUT/probe_${W}_b.log 2>&1
n=c; timeout 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE -d $OUT/probe_${W}_c -o p -- $B > $OUT/probe_${W}_c.log 2>&1
cd $REPO
P=$(find $OUT/probe_${W}_a $OUT/probe_${W}_b $OUT/probe_${W}_c -name '*.db')
[ -n "$P" ] && python tools/rocpd_summary.py pmc $P > $OUT/pmc_probe_$W.txt
find $OUT -name '*.db' -size +8M -delete
grep -vE "k_gen_base|batch_invert|keypair|sign_" $OUT/pmc_probe_$W.txt
tail -3 $OUT/probe_${W}_c.log
