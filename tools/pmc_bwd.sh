# PMC passes over single weight-gradient / backward-data launches (tools/bench_bwd.py --label ...):  LABEL=L1.q.s1 ONLY=wgrad
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
CMD="python $R/tools/bench_bwd.py --only ${ONLY:-wgrad} --label ${LABEL:-L1.q.s1} --reps 5"
cd /tmp
rm -rf $R/gpurun_out/pmc_bwd*
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bwd1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bwd2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bwd3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bwd4 -- $CMD > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
summary = {}
for d in ('pmc_bwd1', 'pmc_bwd2', 'pmc_bwd3', 'pmc_bwd4'):
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for p in glob.glob('gpurun_out/%s/**/*counter_collection*.csv' % d, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if 'wgrad' in n or 'conv_mfma' in n:
                dd = acc[n.replace('void (anonymous namespace)::', '')[:44]][r['Counter_Name']]
                dd[r['Dispatch_Id']] = dd.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    for k, cc in acc.items():
        summary.setdefault(k, {}).update({c: round(sum(v.values()) / len(v)) for c, v in cc.items()})
for k, r in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r:
        r['mfma_pipe_utilisation'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['GRBM_GUI_ACTIVE'] / 8.0), 3)
    print(k, r)
json.dump(summary, open('gpurun_out/pmc_bwd.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_bwd1 gpurun_out/pmc_bwd2 gpurun_out/pmc_bwd3 gpurun_out/pmc_bwd4
