#!/bin/bash
# generic GPU-box session: TESTS (pytest paths, optional), then every command in CMDS (newline separated), logs under gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export NLT_PARITY_DUMP="$GRAFT_REPO_ROOT/gpurun_out/parity_sizes.json"
if [ -n "${TESTS:-}" ]; then
  (timeout ${TMO:-1500} python -m pytest $TESTS -m gpu -q --maxfail=20 --tb=short --timeout=900 -p no:cacheprovider 2>&1 | tail -120) > gpurun_out/pytest_gpu.log
  tail -50 gpurun_out/pytest_gpu.log
fi
i=0
while IFS= read -r cmd; do
  [ -z "$cmd" ] && continue
  i=$((i+1))
  echo "=== $cmd"
  (timeout 600 bash -c "$cmd" 2>&1 | tail -${TAIL:-60}) | tee gpurun_out/cmd_$i.log
done <<< "${CMDS:-}"
