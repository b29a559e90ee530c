#!/bin/bash
# tools/pmc_cache.sh [OP [KERNEL-PATTERN]] -- L2 (TCC) and L1 (TCP) counters of one pass (default: verify), one rocprofv3 --pmc pass
# per counter group (no other tracing in the same run), summarised per kernel.  -> gpurun_out/pmc_cache_OP/, profiles/rNN_pmc_cache*.txt
OP=${1:-verify}; PAT=${2:-verify_fast}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_cache_$OP; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/ab_bench.py $R/curve25519_amd/libcurve25519_amd.so --ops $OP --rounds 1"
i=0
for grp in "TCC_HIT TCC_MISS TCC_REQ TCC_READ" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ" "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES" "TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_LEVEL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d $O/g$i -o c -- $CMD > $O/g$i.log 2>&1; echo "group $i rc=$?"
done
P=$(find $O -name '*.db')
python $R/tools/rocpd_summary.py pmc $P | grep -E "^kernel|$PAT" | tee $O/summary.txt
find $O -name '*.db' -size +8M -delete
