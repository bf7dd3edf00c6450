# rocprofv3 kernel trace of the headline forward -> per-step timeline summary (overlap of the two streams, idle gaps)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf "$R/gpurun_out/prof_fwd"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_fwd" -o fwd -- python $R/bench.py --steps 12 --warmup 3 --headline-only --tune-cache $R/gpurun_out/tune_fused.json > "$R/gpurun_out/prof_fwd_bench.json" 2> "$R/gpurun_out/prof_fwd.err"
cd $R; tail -1 gpurun_out/prof_fwd.err; python tools/train_timeline.py gpurun_out/prof_fwd gpurun_out/fwd_timeline.txt 8 warp_kernel
find gpurun_out/prof_fwd -name "*.csv" -size +4M -delete
