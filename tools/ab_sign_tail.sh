#!/bin/bash
# tools/ab_sign_tail.sh -- the signing pass with its shipped tail (k_batch_invert<FinishPack> + k_ed25519_sign_finish) against
# the one-launch tail k_ed25519_sign_tail<256 / 512 / 1024> (C25519_AMD_SIGN_TAIL), interleaved in one process, outputs compared;
# then the per-kernel times of each under rocprofv3.  -> profiles/rNN_ab_sign_tail.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/curve25519_amd/libcurve25519_amd.so
python $R/tools/ab_bench.py $L $L@C25519_AMD_SIGN_TAIL=256 $L@C25519_AMD_SIGN_TAIL=512 $L@C25519_AMD_SIGN_TAIL=1024 --ops sign --rounds 5 2>&1 | grep -E "^sign|all ok|differ"
cd /tmp && export TMPDIR=/tmp
for t in 0 256 512 1024; do
  mkdir -p $R/gpurun_out/abst/$t
  C25519_AMD_SIGN_TAIL=$t timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abst/$t -o s -- python $R/tools/ab_bench.py $L --ops sign --rounds 3 > $R/gpurun_out/abst/$t.log 2>&1
  S=$(find $R/gpurun_out/abst/$t -name "s_results.db" | head -1)
  echo "== C25519_AMD_SIGN_TAIL=$t"; python $R/tools/rocpd_summary.py stats $S | grep -E "^k_|^void k_" | grep -E "sign|invert" | cut -c1-150
done
