#!/bin/bash
# tools/r04_probe.sh -- one gpurun call: in-kernel cycle probe of the X25519 kernels (raw stamps kept for offline analysis)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${ROUND:-r04a}; mkdir -p $O; cd $R
timeout 600 python tools/cycle_probe.py build_ab/probe1.so --dump $O/probe_fused.npz > $O/cycle_probe_fused.txt 2>&1; echo "probe rc=$?"; cat $O/cycle_probe_fused.txt
timeout 600 python tools/cycle_probe.py build_ab/probe1.so --split --dump $O/probe_split.npz > $O/cycle_probe_split.txt 2>&1; echo "probe rc=$?"; cat $O/cycle_probe_split.txt
