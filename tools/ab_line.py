"""stdin: bench.py output -> 'value ms_per_step [train ms]' of its JSON line (A/B sessions on the GPU box)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d.get('train_step') or {}
print(d['value'], d['ms_per_step'], t.get('ms_per_step'), (d.get('roofline') or {}).get('frac'),
      {k: v['ms_per_step'] for k, v in (d.get('pipelined') or {}).get('lanes', {}).items()})
