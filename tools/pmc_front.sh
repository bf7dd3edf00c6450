# PMC pass over the bench forward: SQ counters of the front kernel (and everything else in the step)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
python bench.py --steps 5 --warmup 2 --headline-only --tune-cache gpurun_out/tune_fused.json > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --headline-only --tune-cache $R/gpurun_out/tune_fused.json > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python $R/bench.py --steps 3 --warmup 1 --headline-only --tune-cache $R/gpurun_out/tune_fused.json > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq3 -- python $R/bench.py --steps 3 --warmup 1 --headline-only --tune-cache $R/gpurun_out/tune_fused.json > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq4 -- python $R/bench.py --steps 3 --warmup 1 --headline-only --tune-cache $R/gpurun_out/tune_fused.json > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
summary = {}
for d in ('pmc_sq', 'pmc_sq2', 'pmc_sq3', 'pmc_sq4'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob('gpurun_out/%s/**/*counter_collection*.csv' % d, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if "front4_kernel" in n or "front_kernel" in n or "back_kernel" in n or "conv_tile" in n or "warp_kernel" in n or "dec_block" in n:
                acc[n[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in acc.items():
        row = {c: round(sum(x)/len(x)) for c, x in dd.items()}
        summary.setdefault(k, {}).update(row)
        print(d, k, row)
for k, r in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
        r['mfma_pipe_utilisation'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['GRBM_GUI_ACTIVE'] / 8.0), 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'SQ_BUSY_CYCLES' in r:
        # same counter pass: MFMA busy cycles per SIMD / SQ busy cycles per shader engine (32 of them) -- independent of what else
        # the GRBM pass saw; the GRBM-based figure above is only meaningful when that pass profiled the same dispatches
        r['mfma_busy_over_sq_busy'] = round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (r['SQ_BUSY_CYCLES'] / 32.0), 3)
json.dump(summary, open('gpurun_out/pmc_sq.json', 'w'), indent=1)
PY
