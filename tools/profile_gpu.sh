#!/bin/bash
# tools/profile_gpu.sh -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then
# separate PMC passes (never combined with tracing other than --kernel-trace) for HBM traffic and VALU use.
# Results land in gpurun_out/prof_*; the summaries worth keeping are copied into profiles/ by hand.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- $BENCH > $OUT/prof_stats.log 2>&1
BENCHX="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu"
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $BENCHX > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $BENCHX > $OUT/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/prof_sq -o sq -- $BENCHX > $OUT/prof_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM -d $OUT/prof_sq2 -o sq2 -- $BENCHX > $OUT/prof_sq2.log 2>&1
find $OUT -name '*.csv' | head -50
for f in $(find $OUT/prof_stats -name '*kernel_stats*.csv'); do echo "== $f"; cat $f; done
