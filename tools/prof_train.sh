cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
cd /tmp && rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train" -o train -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 6 --tune-cache $R/gpurun_out/tune_fused.json > /dev/null 2>&1
cd $R; db=$(find gpurun_out/prof_train -name "*.db" | head -1); python tools/rocpd_summary.py "$db" gpurun_out/train_kernel_stats.csv; cut -c1-130 gpurun_out/train_kernel_stats.csv | head -30; rm -f $db
