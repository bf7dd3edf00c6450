"""stdin/argv: bench.py JSON -> the pipelined sub-lines (headline + released shapes)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d.get('pipelined', {}).get('lanes', {}).items()})
rs = d.get('released_shapes') or {}
for k, v in (rs.get('forward_4_frames_pipelined') or {}).items():
    if isinstance(v, dict):
        print(k, {a: b['ms_per_step'] for a, b in v.items()})
c = d.get('config5_2048_bf16')
if c:
    print('config5 fp32', c['fp32']['ms_per_step'], 'bf16', c['bf16']['ms_per_step'])
