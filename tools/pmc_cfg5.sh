# FETCH_SIZE / WRITE_SIZE passes over the forward of BASELINE config 5 (2048^2 UV, 2 frames, k = 1), fp32 and bf16 middle:
# the whole-pass HBM bytes bench.py's `config5_2048_bf16` sub-line divides by its step time (profiles/*_pmc_traffic_cfg5_<prec>.json)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for P in fp32 bf16; do
  cd /tmp
  NLT_PRECISION=$P timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc5_fetch" -- python $R/tools/bench_cfg.py 256 2048 2 1 512 quiet > "$R/gpurun_out/pmc5_$P.log" 2>&1
  NLT_PRECISION=$P timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc5_write" -- python $R/tools/bench_cfg.py 256 2048 2 1 512 quiet > /dev/null 2>&1
  cd $R; grep "ms / step" gpurun_out/pmc5_$P.log
  python tools/pmc_summary.py gpurun_out/pmc_traffic_cfg5_$P.json gpurun_out/pmc5_fetch gpurun_out/pmc5_write | tail -3
  rm -rf gpurun_out/pmc5_fetch gpurun_out/pmc5_write
done
