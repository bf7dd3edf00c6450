cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/bench_conv.py --preset L2.o.s1,L2.o.s2,L3.o.s1,L3.o.s2,L4.o.s1,L4.o.s2,L5.o.s1,L5.o.s2,L6.o.s1,L6.q.s2,L4.q.s1,L4.q.s2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bench.txt
cd /tmp
for v in lds32f lds64u; do
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_conv_$v -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --preset L4.o.s1 --only $v --reps 3 > /dev/null 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for v in ('lds32f','lds64u'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob('gpurun_out/pmc_conv_%s/**/*counter_collection*.csv' % v, recursive=True):
        for r in csv.DictReader(open(p)):
            if 'conv_tile' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(v, k, {c: round(sum(x)/len(x)) for c, x in d.items()})
PY
