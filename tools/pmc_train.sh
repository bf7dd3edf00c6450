# PMC passes over the train step's fused-end kernels (front_kernel<false>, back_kernel, back_bwd_kernel, front_bwd_kernel, level split)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-released-shapes --train-steps 4 --train-loss l2 --tune-cache $R/gpurun_out/tune_fused.json"
$CMD > /dev/null 2>&1
cd /tmp
rm -rf $R/gpurun_out/pmc_tr*
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr4 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr5 -- $CMD > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
summary = {}
want = ('front_kernel', 'back_kernel', 'back_bwd_kernel', 'front_bwd_kernel', 'level_split', 'warp')
for d in ('pmc_tr1', 'pmc_tr2', 'pmc_tr3', 'pmc_tr4', 'pmc_tr5'):
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for p in glob.glob('gpurun_out/%s/**/*counter_collection*.csv' % d, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if any(w in n for w in want):
                dd = acc[n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]][r['Counter_Name']]
                dd[r['Dispatch_Id']] = dd.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    for k, cc in acc.items():
        summary.setdefault(k, {}).update({c: round(sum(v.values()) / len(v)) for c, v in cc.items()})
for k, r in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r:
        r['mfma_pipe_utilisation'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['GRBM_GUI_ACTIVE'] / 8.0), 3)
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_MB'] = round((2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024 / 1e6, 1)
    print(k, r)
json.dump(summary, open('gpurun_out/pmc_train.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_tr1 gpurun_out/pmc_tr2 gpurun_out/pmc_tr3 gpurun_out/pmc_tr4 gpurun_out/pmc_tr5
