cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_q -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
