"""Regenerate the ablation sources used by run_diag.py / run_stem_diag.py from the product kernels: the conv LDS-DMA kernel
and the fused stem with compile-time hooks (-DDIAG_NOSTORE, -DDIAG_NOFLUSH, -DDIAG_NODMA, -DDIAG_NOPOOL, -DDIAG_ONESTEP)
that remove one cost at a time.  Build each variant with
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I../../squeezedet-pytorch_amd/csrc \
          -D<HOOK> -shared -o libdiag_<name>.so conv_diag.hip        (libstem_<name>.so from stem_diag.hip)
The generated .hip / .so / .s files are scratch (git-ignored)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', '..', 'squeezedet-pytorch_amd', 'csrc')

src = open(os.path.join(csrc, 'conv_igemm.hip')).read()
a = src.index("__global__ __launch_bounds__(WM * 64, MINW) void conv_dma_kernel")
b = src.index("static int sqd_num_cus()")
k = src[a:b]
st = "          *(f32x4*)(ybase + off) = v;\n"
assert st in k
k = k.replace(st, "#ifndef DIAG_NOSTORE\n" + st + "#else\n          if (v.x == 123.456f) *(f32x4*)(ybase + off) = v;\n#endif\n")
fl = "      if (pending) { flush(ptp); pending = false; }\n"
assert fl in k
k = k.replace(fl, "#ifdef DIAG_NOFLUSH\n      if (pending) { if (outv[0][0].x == 123.456f) flush(ptp); pending = false; }\n#else\n" + fl + "#endif\n")
hn = "      const int has_next = (last_i ^ 1) | more_i;\n"
assert hn in k
k = k.replace(hn, "#ifdef DIAG_NODMA\n      const int has_next = 0;\n#else\n" + hn + "#endif\n")
open(os.path.join(here, 'conv_diag.hip'), 'w').write(src[:a] + k + src[b:])

s = open(os.path.join(csrc, 'stem_pool.hip')).read()
s = s.replace("    if (has_next && is_prod) dma_in(nxt, buf ^ 1);", "#ifndef DIAG_NODMA\n    if (has_next && is_prod) dma_in(nxt, buf ^ 1);\n#endif")
s = s.replace("      const int key = is_pool ? pl_key[it] : -1;", "#ifdef DIAG_NOPOOL\n      const int key = (cur.ty == 12345) ? pl_key[it] : -1;\n#else\n      const int key = is_pool ? pl_key[it] : -1;\n#endif")
s = s.replace("        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j], bf[i], acc[i][j]);\n    }\n    // bias + ReLU",
              "        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(af[j], bf[i], acc[i][j]);\n#ifdef DIAG_ONESTEP\n      if (cur.ty != 12345) break;\n#endif\n    }\n    // bias + ReLU")
open(os.path.join(here, 'stem_diag.hip'), 'w').write(s)
print('wrote conv_diag.hip, stem_diag.hip')
