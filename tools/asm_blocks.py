"""Per basic block of a gfx950 ISA listing: instruction counts by kind (MFMA, VALU, lane moves of SGPR spills, LDS, VMEM, scratch).
usage: python tools/asm_blocks.py file.s [min_mfma]   -- prints the blocks holding at least min_mfma MFMAs (default 1)"""
import re
import sys


def blocks(path):
    out, cur, name = [], None, None
    for line in open(path):
        s = line.strip()
        m = re.match(r'^(\.LBB\d+_\d+|_Z\w+):', s)
        if m:
            cur = {'label': m.group(1), 'mfma': 0, 'valu': 0, 'lane': 0, 'ds': 0, 'vmem': 0, 'scratch': 0, 'salu': 0, 'n': 0}
            out.append(cur)
            continue
        if cur is None or not s or s.startswith(('.', ';')):
            continue
        op = s.split()[0]
        cur['n'] += 1
        if op.startswith('v_mfma'):
            cur['mfma'] += 1
        elif op in ('v_readlane_b32', 'v_writelane_b32'):
            cur['lane'] += 1
        elif op.startswith('v_'):
            cur['valu'] += 1
        elif op.startswith('ds_'):
            cur['ds'] += 1
        elif op.startswith('scratch_'):
            cur['scratch'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_')):
            cur['vmem'] += 1
        elif op.startswith('s_'):
            cur['salu'] += 1
    return out


if __name__ == '__main__':
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for b in blocks(sys.argv[1]):
        if b['mfma'] >= lim:
            print(b)
