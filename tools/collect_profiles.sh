#!/bin/bash
# tools/collect_profiles.sh SRC [ROUND] -- copy what tools/gpu_round.sh left under gpurun_out/SRC into profiles/ROUND_* (paths
# of the GPU box stripped, the loader's amdgpu.ids complaint dropped); only files that exist in SRC are replaced.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/$1; R=${2:-r06}; P=profiles
strip() { grep -v "amdgpu.ids" "$1" | sed 's#/tmp/code/msotoodeh__curve25519/repo/##g; s#/root/repo/##g'; }
for f in bench bench_mixed bench_dist1 pmc cycle_probe mad_peak mad_cycles valu_rates; do [ -f $O/$f.json ] && cp $O/$f.json $P/${R}_$f.json; done
for f in kernel_stats pmc mad_peak valu_rates field_ab; do [ -f $O/$f.txt ] && strip $O/$f.txt > $P/${R}_$f.txt; done
[ -f $O/hostapi_rate.txt ] && strip $O/hostapi_rate.txt | grep -v -E "^(RCCL|HIP|ROCm) version|^Hostname|^Librccl" > $P/${R}_hostapi.txt
if [ -f $O/cycle_probe.txt ]; then
  { strip $O/cycle_probe.txt; [ -f $O/cycle_probe_sections.txt ] && strip $O/cycle_probe_sections.txt | sed -n '/== sections/,$p'; } > $P/${R}_cycle_probe.txt
fi
for f in single_call small_batch_sweep long_differential; do
  if [ -f $O/$f.txt ]; then { grep "^#" $P/${R}_$f.txt 2>/dev/null; strip $O/$f.txt | grep -v "^#"; } > /tmp/cp_$f.txt; mv /tmp/cp_$f.txt $P/${R}_$f.txt; fi
done
[ -f $O/batch_sweep.txt ] && { echo "# tools/batch_sweep.sh: N, operation, min ms per call, M ops/s (tools/ab_bench.py at the sustained clock, HBM-resident)"; strip $O/batch_sweep.txt | grep -v "verify True"; } > $P/${R}_batch_sweep.txt
git status --short $P | head -30
