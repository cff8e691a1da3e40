"""Build-time guard (CPU tier, needs hipcc): the wave-autonomous stem kernels retire their prefetch with COUNTED ``s_waitcnt vmcnt``
waits, and a scratch reload is a vector-memory operation that completes in order behind that prefetch -- so a register spill in
their loops costs a memory round trip per tile (round 3 measured 1.5 ms instead of 0.1 ms for the gather kernel, and no gain at all
for the arg-max stem).  Register allocation there proved fragile (one extra ``select`` tipped it into ~600 spilled registers), so the
shipped instantiations are compiled to ISA with the Makefile's flags and their metadata must report zero spills."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'squeezedet-pytorch_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
BASE = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + os.path.join(HERE, '..', 'include'), '--cuda-device-only', '-S']


def _spills(src, extra):
    out = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'sqd_spill_{os.path.basename(src)}.s')
    subprocess.run([HIPCC] + BASE + extra + [os.path.join(CSRC, src), '-o', out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.remove(out)
    res = {}
    for m in re.finditer(r'\.name:\s+(\S+)(.*?)\.vgpr_spill_count:\s+(\d+)', text, flags=re.S):
        res[m.group(1)] = int(m.group(3))
    return res


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_wave_autonomous_kernels_do_not_spill():
    g = _spills('stem_wgrad_gather.hip', ['-fno-slp-vectorize'])           # (the Makefile builds this file without the SLP vectoriser)
    shipped = [k for k in g if 'stem_wgrad_gather_kernelILb0ELi2E' in k]
    assert shipped and all(g[k] == 0 for k in shipped), {k: g[k] for k in shipped}
    s = _spills('stem_pool.hip', [])
    want = ['stem_wave_kernelILi2ELi2ELb0ELi0E',       # inference stem
            'stem_wave_kernelILi2ELi2ELb0ELi16E',      # inference stem + the first Fire's squeeze
            'stem_wave_kernelILi1ELi2ELb1ELi0E']       # training stem (arg-max codes)
    for w in want:
        hits = [k for k in s if w in k]
        assert hits, (w, sorted(s)[:8])
        assert all(s[k] == 0 for k in hits), {k: s[k] for k in hits}
