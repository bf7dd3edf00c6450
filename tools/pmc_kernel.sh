# SQ counter passes over any command for kernels whose name contains $KERNEL:  CMD="python tools/bench_front_ovr.py" KERNEL=front_ovr bash tools/pmc_kernel.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; OUT=${OUT:-pmc_kernel}
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${OUT}_$i -- bash -c "cd $R && $CMD" > /dev/null 2>&1
done
cd $R
KERNEL="$KERNEL" OUT="$OUT" python - <<'PY'
import csv, glob, collections, json, os
K, OUT = os.environ['KERNEL'], os.environ['OUT']
summary = collections.defaultdict(dict)
for i in range(1, 7):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for p in glob.glob('gpurun_out/%s_%d/**/*counter_collection*.csv' % (OUT, i), recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if K in n:
                acc[n[:60]][r['Counter_Name']][r.get('Dispatch_Id')] += float(r['Counter_Value'])
    for k, dd in acc.items():
        summary[k].update({c: round(sum(x.values()) / len(x)) for c, x in dd.items()})
for k, r in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'SQ_BUSY_CYCLES' in r:
        r['mfma_busy_over_sq_busy'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['SQ_BUSY_CYCLES'] / 32.0), 3)
    if 'SQ_WAIT_ANY' in r and 'SQ_WAVE_CYCLES' in r:
        r['wait_any_over_wave_cycles'] = round(r['SQ_WAIT_ANY'] / r['SQ_WAVE_CYCLES'], 3)
    if 'SQ_ACTIVE_INST_VALU' in r and 'SQ_WAVE_CYCLES' in r:
        r['valu_active_over_wave_cycles'] = round(r['SQ_ACTIVE_INST_VALU'] / r['SQ_WAVE_CYCLES'], 3)
    print(k, json.dumps(r, indent=1))
json.dump(summary, open('gpurun_out/%s.json' % OUT, 'w'), indent=1)
PY
