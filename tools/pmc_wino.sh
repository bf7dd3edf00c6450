# SQ counter passes over one stride-1 conv shape on the direct, three-term and Winograd kernels (tools/run_wino_shape.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
SHAPE="${SHAPE:-64 128 4 4}"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w1 -- python $R/tools/run_wino_shape.py $SHAPE > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w2 -- python $R/tools/run_wino_shape.py $SHAPE > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w3 -- python $R/tools/run_wino_shape.py $SHAPE > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
summary = {}
for d in ('pmc_w1', 'pmc_w2', 'pmc_w3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for p in glob.glob('gpurun_out/%s/**/*counter_collection*.csv' % d, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if 'conv_' in n:
                acc[n.replace('void (anonymous namespace)::', '')[:60]][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, dd in acc.items():
        row = {c: round(sum(x.values()) / len(x)) for c, x in dd.items()}
        summary.setdefault(k, {}).update(row)
for k, r in summary.items():
    wc = r.get('SQ_WAVE_CYCLES', 0)
    if wc:
        r['frac_wait_any(waitcnt/barrier)'] = round(r.get('SQ_WAIT_ANY', 0) / wc, 3)
        r['frac_wait_inst(issue stall)'] = round(r.get('SQ_WAIT_INST_ANY', 0) / wc, 3)
        r['frac_active'] = round(r.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'SQ_BUSY_CYCLES' in r:
        r['mfma_busy_over_sq_busy'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['SQ_BUSY_CYCLES'] / 32.0), 3)
    print(k, json.dumps(r))
json.dump(summary, open('gpurun_out/pmc_wino.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_w1 gpurun_out/pmc_w2 gpurun_out/pmc_w3
