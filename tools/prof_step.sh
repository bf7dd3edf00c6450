# rocprofv3 kernel trace of the headline forward at precision $PREC -> every kernel of the last step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for P in ${PRECS:-f32x3_9 fp32}; do
  rm -rf "$R/gpurun_out/prof_step"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_step" -o fwd -- python $R/bench.py --precision $P --steps 12 --warmup 3 --headline-only > "$R/gpurun_out/prof_step_$P.json" 2> "$R/gpurun_out/prof_step.err"
  cd $R; python tools/step_kernels.py gpurun_out/prof_step > gpurun_out/step_kernels_$P.txt; tail -1 gpurun_out/step_kernels_$P.txt
  rm -rf gpurun_out/prof_step
done
