# PMC pass over the bench forward: L2 (TCC) hits / misses / memory-side read requests per kernel -- which launches re-fetch
# (r04 review: warp_kernel 258 MB vs 84 MB algorithmic, conv_tile3<1,4,9,2> 341 MB vs 201 MB)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
B="python $R/bench.py --steps 3 --warmup 1 --headline-only --tune-cache $R/gpurun_out/tune_fused.json"
python bench.py --steps 5 --warmup 2 --headline-only --tune-cache gpurun_out/tune_fused.json > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tcc1 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tcc2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tcc3 -- $B > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
summary = collections.defaultdict(dict)
for d in ('pmc_tcc1', 'pmc_tcc2', 'pmc_tcc3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob('gpurun_out/%s/**/*counter_collection*.csv' % d, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, dd in acc.items():
        summary[k].update({c: round(sum(x) / len(x)) for c, x in dd.items()})
        summary[k]['calls_' + d] = max(len(x) for x in dd.values())
out = {}
for k, r in sorted(summary.items(), key=lambda kv: -kv[1].get('TCC_REQ_sum', 0)):
    if 'TCC_HIT_sum' in r and r['TCC_HIT_sum'] + r['TCC_MISS_sum'] > 0:
        r['l2_hit_rate'] = round(r['TCC_HIT_sum'] / (r['TCC_HIT_sum'] + r['TCC_MISS_sum']), 4)
    if 'TCC_EA0_RDREQ_sum' in r:
        r32 = r.get('TCC_EA0_RDREQ_32B_sum', 0)
        r['ea_read_bytes_if_64B'] = int(64 * (r['TCC_EA0_RDREQ_sum'] - r32) + 32 * r32)
        r['ea_read_bytes_if_128B'] = int(128 * (r['TCC_EA0_RDREQ_sum'] - r32) + 32 * r32)
    out[k] = r
json.dump(out, open('gpurun_out/pmc_tcc.json', 'w'), indent=1)
for k, r in list(out.items())[:24]:
    print(k[:60], {a: b for a, b in r.items() if not a.startswith('calls_')})
PY
find gpurun_out/pmc_tcc1 gpurun_out/pmc_tcc2 gpurun_out/pmc_tcc3 -name "*.csv" -size +4M -delete
