#!/bin/bash
# tools/build_variants.sh NAME "FLAGS" [NAME "FLAGS" ...] -- A/B builds of the engine into build_ab/NAME.so
# (timed against each other by tools/ab_bench.py, which also checks that every build produces the same bytes)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=131072 $flags curve25519_amd/csrc/engine.hip -o build_ab/$name.so \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | paste - - - - | \
      sed -E 's/.*Function Name: ([^ ]+).*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*/\1 vgpr=\2 scratch=\3 occ=\4/' \
      | grep -E "verify|sign_mult|x25519_fused" > build_ab/$name.txt; echo "built $name ($flags)" ) &
done
wait
