#!/bin/bash
# tools/r04_probe.sh -- one gpurun call: the in-kernel cycle probe of the X25519 kernels (shipped two-launch shape, fused
# kernel, sections of a step), raw stamps kept for offline analysis, then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${ROUND:-r04c}; mkdir -p $O; cd $R
P=curve25519_amd/libcurve25519_amd_probe.so
timeout 600 python tools/cycle_probe.py $P --dump $O/probe_split.npz --json $O/cycle_probe.json > $O/cycle_probe_split.txt 2>&1; echo "probe rc=$?"
timeout 600 python tools/cycle_probe.py $P --fused --dump $O/probe_fused.npz --sections build_ab/probe2.so build_ab/probe2.s > $O/cycle_probe_fused.txt 2>&1; echo "probe rc=$?"
cat $O/cycle_probe_split.txt $O/cycle_probe_fused.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; tail -3 $O/bench.err
