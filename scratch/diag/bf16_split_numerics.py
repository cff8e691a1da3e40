"""VERDICT round 4, item 6 (experiment, report only): fp32 = three bf16 pieces exactly, so a Winograd-domain dot product over C channels can run
on the bf16 matrix cores as 9 (exact products, fp32 accumulate) or 6 (cross terms below 2^-24 dropped) bf16 MFMA products per fp32 product.
This script measures the NUMERICS on the CPU (numpy, no GPU): ConvDet-shaped reductions (C = 768, transformed ReLU activations x
transformed N(0, 0.002) .. Kaiming weights) against a float64 reference, next to the fp32 fma chain the shipped kernels execute
(v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered fma chain, scratch/probe.py).

Emulation: bf16 = round-to-nearest-even on the upper 16 bits; a bf16 x bf16 product is exact in fp32; one v_mfma_f32_16x16x32_bf16 adds 32 such
products to its fp32 accumulator -- its internal summation order / width is not documented, so two bounds are reported: 'wide' (the 32
products summed exactly, ONE rounding per MFMA) and 'narrow' (a sequential fp32 chain over the 32 products).  Piece products are issued
smallest first (a2b2 ... a0b0) into the same accumulator.
usage: python scratch/diag/bf16_split_numerics.py"""
import numpy as np

def bf16_round(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    u = (u + 0x7fff + lsb) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)

def split3(x):
    x = np.asarray(x, np.float32)
    a0 = bf16_round(x); r = (x - a0).astype(np.float32)
    a1 = bf16_round(r); r2 = (r - a1).astype(np.float32)
    a2 = bf16_round(r2)
    return a0, a1, a2

def fma_chain(u, v):
    """fp32 k-ordered fma chain over the last axis (what v_mfma_f32_16x16x4_f32 does): acc = fma(u_k, v_k, acc)."""
    acc = np.zeros(u.shape[:-1], np.float32)
    for k in range(u.shape[-1]):
        acc = (acc.astype(np.float64) + u[..., k].astype(np.float64) * v[..., k].astype(np.float64)).astype(np.float32)   # fma: one rounding
    return acc

def bf16_products(u, v, pairs, wide):
    us, vs = split3(u), split3(v)
    acc = np.zeros(u.shape[:-1], np.float32)
    C = u.shape[-1]
    for kb in range(0, C, 32):
        for (i, j) in pairs:                                   # smallest terms first
            p = us[i][..., kb:kb + 32].astype(np.float64) * vs[j][..., kb:kb + 32].astype(np.float64)     # exact products
            if wide:
                acc = (acc.astype(np.float64) + p.sum(-1)).astype(np.float32)
            else:
                for k in range(p.shape[-1]):
                    acc = (acc + p[..., k].astype(np.float32)).astype(np.float32)
    return acc

P9 = [(2, 2), (2, 1), (1, 2), (2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
P6 = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
P3 = [(1, 0), (0, 1), (0, 0)]

rs = np.random.RandomState(0)
C, NROW = 768, 4096
# transformed activations: B^T d B of ReLU(N(0,1)) patches -> sums / differences of 4 values; transformed weights: G g G^T of N(0, s)
d = np.maximum(rs.standard_normal((NROW, C, 4, 4)), 0).astype(np.float32)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
V = np.einsum('ij,ncjk,lk->ncil', Bt, d, Bt).astype(np.float32)
g = (rs.standard_normal((NROW, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
U = np.einsum('ij,ncjk,lk->ncil', G, g, G).astype(np.float32)
print(f'ConvDet-shaped Winograd-domain dot products: C = {C}, {NROW} (tile, channel) rows x 16 positions')
for pos in [(0, 0), (1, 1), (2, 3)]:
    u = U[:, :, pos[0], pos[1]]; v = V[:, :, pos[0], pos[1]]
    ref = (u.astype(np.float64) * v.astype(np.float64)).sum(-1)
    scale = np.abs(ref).max()
    rows = [('fp32 fma chain (shipped kernels)', fma_chain(u, v))]
    for name, pairs in (('bf16 x 9 products', P9), ('bf16 x 6 products', P6), ('bf16 x 3 products', P3)):
        rows.append((name + ', one rounding per MFMA', bf16_products(u, v, pairs, True)))
        rows.append((name + ', fp32 chain inside the MFMA', bf16_products(u, v, pairs, False)))
    print(f' position {pos}: |ref| max {scale:.3f}')
    for name, got in rows:
        e = np.abs(got.astype(np.float64) - ref)
        print(f'   {name:58s} max err {e.max() / scale:.2e}   rms {np.sqrt((e ** 2).mean()) / scale:.2e}   (of max |ref|)')
