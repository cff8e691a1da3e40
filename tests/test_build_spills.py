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


def _spills(src, extra, scratch=False):
    """{kernel: spilled VGPRs} of one source file (``scratch``: {kernel: (spilled VGPRs, bytes of private segment)})."""
    out = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'sqd_spill_{os.path.basename(src)}.s')
    subprocess.run([HIPCC] + BASE + extra + [os.path.join(CSRC, src), '-o', out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.remove(out)
    res = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', text, flags=re.S):
        body = m.group(2)
        sp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', body).group(1))
        pv = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', body).group(1))
        res[m.group(1)] = (sp, pv) if scratch else sp
    return res


# the Makefile's flags for the Winograd translation units (no SI load/store optimizer)
WINO = ['-Xclang', '-target-feature', '-Xclang', '-load-store-opt']


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


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_counted_wait_kernels_of_the_training_step_do_not_spill():
    """Every kernel whose barrier / tile hand-off waits with a COUNTED ``s_waitcnt vmcnt(N)`` over always-issued stores relies on the
    compiler adding no vector-memory instruction of its own (a scratch reload) between those stores and the wait
    (MI355X_MICROARCH.md: loads, stores and LDS-DMA count together, in issue order): the fused squeeze backward
    conv_wgrad_kernel<1, TN, TC, 2, DG = true> (csrc/wgrad.hip), the storing forms of the two Fire bridges and of the stem + squeeze
    launch (wino_bridge.h MODE 2, wino_poolbridge.h SAVE, stem_pool.hip ARGMAX + SQ), and the balanced Winograd kernel's plain
    instantiation.  Zero spilled registers and zero scratch for all of them; the dropout-carrying instantiation of the balanced
    kernel is allowed its 14 documented spills (outside the matrix-core blocks)."""
    w = _spills('wgrad.hip', [], scratch=True)
    dg = [k for k in w if re.search(r'conv_wgrad_kernelILi1ELi\d+ELi\d+ELi2ELb1EE', k)]
    assert len(dg) >= 20, sorted(w)[:6]
    assert all(w[k] == (0, 0) for k in dg), {k: w[k] for k in dg if w[k] != (0, 0)}
    c = _spills('conv_wino.hip', WINO, scratch=True)
    for pat in (r'fire_bridge16_kernelILi\dELi2ELi\dEE', r'fire_bridge16_kernelILi\dELi1ELi\dEE', r'fire_poolbridge16_kernelILi\dELi\dELb[01]EE'):
        hits = [k for k in c if re.search(pat, k)]
        assert len(hits) >= 4, (pat, sorted(c)[:6])
        assert all(c[k] == (0, 0) for k in hits), {k: c[k] for k in hits if c[k] != (0, 0)}
    s = _spills('stem_pool.hip', [], scratch=True)
    hits = [k for k in s if 'stem_wave_kernelILi1ELi2ELb1ELi16E' in k]
    assert hits and all(s[k] == (0, 0) for k in hits), {k: s[k] for k in hits}
    k = _spills('conv_wino_sk.hip', WINO, scratch=True)
    plain = [n for n in k if 'conv_wino_sk_kernelILb0E' in n]
    full = [n for n in k if 'conv_wino_sk_kernelILb1E' in n]
    assert plain and full
    assert all(k[n] == (0, 0) for n in plain), {n: k[n] for n in plain}
    assert all(k[n][0] <= 14 for n in full), {n: k[n] for n in full}
