tools/ubench/valu_rate2 > gpurun_out/r2_valu_rate2.log 2>&1
tools/ab.sh > gpurun_out/r2_ab3.log 2>&1; cat gpurun_out/r2_ab3.log
tools/ab.sh --graph-frames 0 >> gpurun_out/r2_ab3.log 2>&1; tail -1 gpurun_out/r2_ab3.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-frames 0 --no-verify --orbit-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/r2_prof3.json 2>$GRAFT_REPO_ROOT/gpurun_out/r2_prof3.err
cd $GRAFT_REPO_ROOT; find gpurun_out/r2_prof3 -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
cat gpurun_out/r2_valu_rate2.log
