#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+per-op tables), rocprofv3 kernel trace of the forward, and two PMC
# passes (FETCH_SIZE / WRITE_SIZE separately, no tracing domains beside --kernel-trace).
#   TESTS="..." restricts pytest; PMC=0 skips the counter passes; SKIP_TESTS=1 skips pytest.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
export NLT_PARITY_DUMP="$GRAFT_REPO_ROOT/gpurun_out/parity_sizes.json"
R="$GRAFT_REPO_ROOT"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  (timeout 1800 python -m pytest ${TESTS:-tests} -m gpu -q --maxfail=30 --tb=short --timeout=900 -p no:cacheprovider 2>&1 | tail -200) > gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
fi
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log; cat gpurun_out/smoke.log
timeout 900 python bench.py --per-op --per-op-train --tune-cache gpurun_out/tune_fused.json > gpurun_out/bench.json 2> gpurun_out/bench_perop.txt
cat gpurun_out/bench.json; tail -130 gpurun_out/bench_perop.txt
B="python $R/bench.py --steps ${PROF_STEPS:-100} --warmup 10 --headline-only --tune-cache $R/gpurun_out/tune_fused.json"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o bench -- $B > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err"
cd "$R"; cat gpurun_out/prof_bench.json; tail -2 gpurun_out/prof.err
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" gpurun_out/kernel_stats.csv && cut -c1-150 gpurun_out/kernel_stats.csv | head -40
rm -f gpurun_out/prof/*.db gpurun_out/prof/*/*.db 2>/dev/null
if [ "${PMC:-1}" = "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -- $B > /dev/null 2> "$R/gpurun_out/pmc_fetch.err"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -- $B > /dev/null 2> "$R/gpurun_out/pmc_write.err"
  cd "$R"; tail -2 gpurun_out/pmc_fetch.err
  python tools/pmc_summary.py gpurun_out/pmc_traffic.json gpurun_out/pmc_fetch gpurun_out/pmc_write
  find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +8M -delete
fi
if [ "${EXTRAS:-1}" = "1" ]; then
  # timelines of the forward and of the train step (overlap / idle analysis), the adversarial warp map, the other configs
  bash tools/prof_fwd.sh > gpurun_out/fwd_timeline.log 2>&1; head -4 gpurun_out/fwd_timeline.txt
  LOSS=l2 bash tools/prof_train.sh > gpurun_out/train_timeline.log 2>&1; head -4 gpurun_out/train_timeline_l2.txt
  timeout 300 python bench.py --steps 50 --warmup 5 --headline-only --warp random --tune-cache gpurun_out/tune_fused.json > gpurun_out/bench_warp_random.json 2>/dev/null; cut -c1-260 gpurun_out/bench_warp_random.json
  (timeout 300 python tools/bench_cfg.py 1024 256 4 1 256; timeout 300 python tools/bench_cfg.py 256 512 4 1 512) > gpurun_out/other_configs.txt 2>&1; grep "ms / step" gpurun_out/other_configs.txt
  timeout 300 python tools/bench_bwd.py > gpurun_out/bench_bwd.txt 2>&1; tail -3 gpurun_out/bench_bwd.txt
  # the reference's inference mode (nlt_test.infer): per-launch table, two lanes, and a kernel trace of the loop
  (timeout 300 python tools/bench_infer.py 256 1024 4 512; timeout 300 python tools/bench_infer.py 256 1024 4 512 lanes=2 quiet; timeout 300 python tools/bench_infer.py 256 512 4 512) > gpurun_out/infer_mode.txt 2>&1; grep "infer depth" gpurun_out/infer_mode.txt
  rm -rf "$R/gpurun_out/prof_infer"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_infer" -o infer -- python $R/tools/bench_infer.py 256 1024 4 512 quiet > "$R/gpurun_out/prof_infer.log" 2>&1)
  python tools/train_timeline.py gpurun_out/prof_infer gpurun_out/infer_timeline.txt 8 warp_kernel > /dev/null 2>&1; head -30 gpurun_out/infer_timeline.txt
  find gpurun_out/prof_infer -name "*kernel_trace.csv" -size +4M -delete
fi
du -sh gpurun_out
