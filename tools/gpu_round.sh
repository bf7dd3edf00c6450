#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+per-op table), rocprofv3 kernel trace.
# TUNE=1 adds the tile sweep.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider 2>&1 | tail -150) > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log; cat gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --per-op > gpurun_out/bench.json 2> gpurun_out/bench_perop.txt
cat gpurun_out/bench.json; tail -70 gpurun_out/bench_perop.txt
if [ "${TUNE:-0}" = "1" ]; then
  timeout 600 python tools/tune_tiles.py > gpurun_out/tune.txt 2> gpurun_out/tune.err; tail -70 gpurun_out/tune.txt; tail -5 gpurun_out/tune.err
fi
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
cd "$GRAFT_REPO_ROOT"; cat gpurun_out/prof_bench.json; tail -3 gpurun_out/prof.err
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" gpurun_out/kernel_stats.csv && cut -c1-160 gpurun_out/kernel_stats.csv | head -50
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_rocprof.csv
rm -rf gpurun_out/prof/*/*.db 2>/dev/null; du -sh gpurun_out
