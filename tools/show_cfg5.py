import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d['config5_2048_bf16']
print(d['value'], d['ms_per_step'], 'config5 fp32', c['fp32']['ms_per_step'], 'bf16', c['bf16']['ms_per_step'])
