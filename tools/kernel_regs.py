"""Register / LDS / spill counts of every kernel in a gfx950 ISA listing (hipcc --cuda-device-only -S).
usage: python tools/kernel_regs.py file.s [name-substring]"""
import re
import sys


def kernel_regs(path):
    t = open(path).read()
    out = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, flags=re.S):
        body = m.group(2)

        def g(k):
            r = re.search(re.escape(k) + r':\s+(\d+)', body)
            return int(r.group(1)) if r else None
        out[m.group(1)] = {k.strip('.'): g(k) for k in ('.sgpr_count', '.sgpr_spill_count', '.vgpr_count', '.vgpr_spill_count', '.agpr_count',
                                                       '.group_segment_fixed_size', '.private_segment_fixed_size')}
    return out


if __name__ == '__main__':
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, v in kernel_regs(sys.argv[1]).items():
        if sub in name:
            print(name[:90], v)
