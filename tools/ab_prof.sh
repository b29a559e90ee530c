# tools/ab_prof.sh OPS build ... -- per-kernel times (rocprofv3 --kernel-trace --stats) of tools/ab_bench.py --ops OPS for each build_ab/<build>.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OPS=$1; shift
for v in "$@"; do
  mkdir -p $R/gpurun_out/abp/$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abp/$v -o s -- python $R/tools/ab_bench.py $R/build_ab/$v.so --ops $OPS --rounds 3 > $R/gpurun_out/abp/$v.log 2>&1
  S=$(find $R/gpurun_out/abp/$v -name "s_results.db" | head -1)
  echo "== $v"; python $R/tools/rocpd_summary.py stats $S | grep -E "^k_|^void k_" | head -12
done
