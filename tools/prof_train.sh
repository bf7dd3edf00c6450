# rocprofv3 kernel trace of the train step -> per-step timeline summary (tools/train_timeline.py).  LOSS=l2|barron
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf "$R/gpurun_out/prof_train"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_train" -o train -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-released-shapes --train-steps 12 --train-loss ${LOSS:-l2} --tune-cache $R/gpurun_out/tune_fused.json > "$R/gpurun_out/prof_train_bench.json" 2> "$R/gpurun_out/prof_train.err"
cd $R; tail -2 gpurun_out/prof_train.err; python tools/train_timeline.py gpurun_out/prof_train gpurun_out/train_timeline_${LOSS:-l2}.txt 6
find gpurun_out/prof_train -name "*.csv" -size +4M -delete
