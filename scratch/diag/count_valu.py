"""Static VALU-per-MFMA count of a kernel's main loop(s) from the ISA dump: usage count_valu.py file.s mangled_prefix"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(pref) and l.rstrip().endswith(':') or (l.startswith(pref) and ':' in l))
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
body = src[start:end]
# loop ranges: from a label marked as loop header to the last backward branch to it
labels = {l.split(':')[0]: i for i, l in enumerate(body) if l.startswith('.LBB')}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
if not loops:
    print('no loops'); sys.exit()
lo = min(a for a, b in loops); hi = max(b for a, b in loops)
def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return None
for name, (a, b) in (('all loops', (lo, hi)),) + tuple((f'loop@{a}', (a, b)) for a, b in sorted(set(loops))[:6]):
    c = collections.Counter()
    ops = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith(';') or t[0].startswith('.'): continue
        k = classify(t[0])
        if k: c[k] += 1
        if k == 'valu': ops[re.sub(r'_e(32|64)$', '', t[0])] += 1
    if c['mfma']:
        print(f"{name:14s} lines {a}-{b}: mfma {c['mfma']} valu {c['valu']} ({c['valu']/c['mfma']:.2f}/mfma) lds {c['lds']} vmem {c['vmem']} salu {c['salu']} wait {c['wait']}")
        if name == 'all loops': print('   top VALU:', ops.most_common(12))
