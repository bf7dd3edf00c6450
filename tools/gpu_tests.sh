#!/bin/bash
# Parity tests on the GPU box.  TESTS="..." restricts pytest (default: everything marked gpu).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
export NLT_PARITY_DUMP="$GRAFT_REPO_ROOT/gpurun_out/parity_sizes.json"
nproc > gpurun_out/nproc.txt
(timeout ${TMO:-1500} python -m pytest ${TESTS:-tests} -m gpu -q --maxfail=30 --tb=short --timeout=900 -p no:cacheprovider --durations=15 2>&1 | tail -150) > gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
cat gpurun_out/parity_sizes.json 2>/dev/null | head -150
