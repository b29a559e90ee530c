#!/bin/bash
# tools/gpu_round.sh -- one gpurun call's worth of work: the GPU test suite, bench lines, the instruction-rate
# microbenchmark, the host-API rates and the rocprofv3 passes.  Everything lands under gpurun_out/$ROUND/ (default r06).
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tests|bench|ubench|prof|all ...]'
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${ROUND:-r06}
mkdir -p $OUT
cd $REPO
WHAT=${*:-all}
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }

if has tests; then
  timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.full 2>&1
  echo "pytest rc=$?" > $OUT/pytest_gpu.rc                      # pytest's own exit code, not a pipe's
  tail -30 $OUT/pytest_gpu.full > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.rc >> $OUT/pytest_gpu.log; rm -f $OUT/pytest_gpu.full
  grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; cat $OUT/pytest_gpu.rc
fi
if has bench; then
  timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  timeout 300 python bench.py --steps 10 --warmup 3 --workload mixed --no-cpu > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err; echo "mixed rc=$?"
  timeout 300 python bench.py --steps 20 --warmup 5 --dist-selftest --no-side --no-cpu > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "dist1 rc=$?"
  timeout 300 python tools/hostapi_rate.py --json $OUT/hostapi_rate.json > $OUT/hostapi_rate.txt 2>&1; echo "hostapi rc=$?"
  cat $OUT/hostapi_rate.txt
fi
if has ubench; then
  timeout 300 tools/ubench/valu_rates $OUT/valu_rates.json > $OUT/valu_rates.txt 2>&1; echo "ubench rc=$?"
  timeout 300 tools/ubench/mad_peak $OUT/mad_peak.json $OUT/mad_cycles.json > $OUT/mad_peak.txt 2>&1; echo "mad_peak rc=$?"; cat $OUT/mad_peak.txt
  [ -x tools/ubench/field_ab ] && { timeout 300 tools/ubench/field_ab > $OUT/field_ab.txt 2>&1; echo "field_ab rc=$?"; cat $OUT/field_ab.txt; }
fi
if has probe; then                                     # in-kernel s_memtime probe of the X25519 kernels (un-profiled launches)
  P=curve25519_amd/libcurve25519_amd_probe.so
  timeout 300 python tools/cycle_probe.py $P --json $OUT/cycle_probe.json > $OUT/cycle_probe.txt 2>&1; echo "probe rc=$?"
  timeout 300 python tools/cycle_probe.py $P --fused >> $OUT/cycle_probe.txt 2>&1; echo "probe fused rc=$?"
  timeout 300 python tools/single_call_latency.py > $OUT/single_call.txt 2>&1; timeout 300 python tools/single_call_breakdown.py >> $OUT/single_call.txt 2>&1
  timeout 600 python tools/small_batch_sweep.py > $OUT/small_batch_sweep.txt 2>&1; echo "sweep rc=$?"
fi
if has longdiff; then                                  # millions of fresh random operations against the reference's own C code
  timeout 900 python tests/long_differential.py --rounds ${LD_ROUNDS:-16} --seed ${LD_SEED:-51} > $OUT/long_differential.txt 2>&1; echo "longdiff rc=$?"
  tail -3 $OUT/long_differential.txt
fi
if has virt; then
  timeout 300 python tools/multi_virtual_rate.py > $OUT/multi_virtual_q4.txt 2>&1; echo "virt rc=$?"
  GPU_MAX_HW_QUEUES=16 timeout 300 python tools/multi_virtual_rate.py > $OUT/multi_virtual_q16.txt 2>&1; echo "virt16 rc=$?"
fi
if has comb; then                                      # A/B of the fixed-base combs, interleaved in one process (tunable BASE_COMB)
  L=curve25519_amd/libcurve25519_amd.so
  timeout 600 python tools/ab_bench.py $L@BASE_COMB=0 $L@BASE_COMB=1 --ops sign,keypair --rounds ${AB_ROUNDS:-6} > $OUT/ab_base_comb.txt 2>&1
  echo "comb rc=$?"; cat $OUT/ab_base_comb.txt
fi
if has ab && ls build_ab/*.so >/dev/null 2>&1; then
  timeout 900 python tools/ab_bench.py curve25519_amd/libcurve25519_amd.so build_ab/*.so ${AB_EXTRA:-} --ops ${AB_OPS:-x25519,sign,verify,keypair} --rounds ${AB_ROUNDS:-4} > $OUT/ab_bench.txt 2>&1
  echo "ab rc=$?"; cat $OUT/ab_bench.txt
fi
if has stats && ! has prof; then                       # the kernel-trace pass alone (per-kernel times of the bench)
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python $REPO/bench.py --steps 30 --warmup 4 --no-cpu > $OUT/prof_stats.log 2>&1
  cd $REPO
  S=$(find $OUT/prof_stats -name '*.db' | head -1)
  [ -n "$S" ] && python tools/rocpd_summary.py stats $S > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
  find $OUT -name '*.db' -size +8M -delete
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  BENCH="python $REPO/bench.py --steps 60 --warmup 6 --no-cpu"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- $BENCH > $OUT/prof_stats.log 2>&1
  BENCHX="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $BENCHX > $OUT/prof_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $BENCHX > $OUT/prof_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/prof_sq -o sq -- $BENCHX > $OUT/prof_sq.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQC_ICACHE_REQ SQC_ICACHE_MISSES -d $OUT/prof_sq2 -o sq2 -- $BENCHX > $OUT/prof_sq2.log 2>&1
  cd $REPO
  S=$(find $OUT/prof_stats -name '*.db' | head -1)
  [ -n "$S" ] && python tools/rocpd_summary.py stats $S > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
  P=$(find $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2 -name '*.db')
  [ -n "$P" ] && python tools/rocpd_summary.py pmc $P > $OUT/pmc.txt && python tools/rocpd_summary.py pmc-json $P > $OUT/pmc.json
  # keep the databases out of the 64 MiB pull
  find $OUT -name '*.db' -size +8M -delete
fi
echo "gpu_round done: $WHAT"
