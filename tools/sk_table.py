"""Write a copy of tuning.json whose listed Winograd rows ('W:C:N:npix') point at another configuration (default: the balanced
stream-K kernel, tiles.WINO_SK_CFG) -- for whole-step A/B runs:  SQD_TUNING_JSON=<out> python bench.py --layers.
usage: python tools/sk_table.py <out.json> <row>[=cfg] [<row>[=cfg] ...]"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(HERE, '..', 'squeezedet-pytorch_amd', 'tuning.json')
out = sys.argv[1]
t = json.load(open(src))
for spec in sys.argv[2:]:
    row, _, cfg = spec.partition('=')
    cfg = int(cfg) if cfg else 16
    e = t.setdefault(row, {'all': {}, 'us': 0.0})
    e['cfg'] = cfg
    e['us'] = 0.0                     # (rows whose 'us' >= 'direct_us' read as "the direct kernel is faster")
json.dump(t, open(out, 'w'), indent=0, sort_keys=True)
print('wrote', out, len(sys.argv) - 2, 'rows changed')
