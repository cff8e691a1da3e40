"""Report only: fp32 Winograd F(4x4,3x3) against the shipped F(2x2,3x3) arithmetic and the direct fp32 convolution, all three against a
float64 truth, on the layer shapes of the SqueezeDet hot path (numpy on the CPU; the position GEMMs in fp32 like the matrix cores).
F(4x4,3x3) executes 36 multiplies per 16 outputs (4x fewer than direct, 1.78x fewer than F(2x2,3x3)) at transforms with entries up to 8
and 1/24: the question is what that costs in error relative to the 1e-4 parity bound."""
import numpy as np
rng = np.random.default_rng(0)
f = np.float32


def direct(x, w, dt):
    B, H, W, C = x.shape; N = w.shape[0]
    xp = np.zeros((B, H + 2, W + 2, C), dt); xp[:, 1:-1, 1:-1] = x
    y = np.zeros((B, H, W, N), dt)
    for r in range(3):
        for s in range(3):
            y += (xp[:, r:r + H, s:s + W, :].reshape(-1, C) @ w[:, :, r, s].T.astype(dt)).reshape(B, H, W, N)
    return y


F23 = (np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], f),
       np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], f),
       np.array([[1, 1, 1, 0], [0, 1, -1, -1]], f), 2)
F43 = (np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
       np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], f),
       np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], f), 4)


def wino(x, w, form):
    G, Bt, At, m = form
    a = m + 2
    B, H, W, C = x.shape; N = w.shape[0]
    U = np.einsum('ir,ncrs,js->ijcn', G.astype(np.float64), w.astype(np.float64), G.astype(np.float64)).astype(f)     # weights transformed once, in float64, rounded to fp32
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = np.zeros((B, Hp + 2, Wp + 2, C), f); xp[:, 1:H + 1, 1:W + 1] = x
    y = np.zeros((B, Hp, Wp, N), f)
    for ty in range(Hp // m):
        for tx in range(Wp // m):
            dd = xp[:, m * ty:m * ty + a, m * tx:m * tx + a, :]
            t = np.einsum('ia,bakc->bikc', Bt, dd).astype(f)
            V = np.einsum('bikc,jk->bijc', t, Bt).astype(f)
            M = np.zeros((B, a, a, N), f)
            for i in range(a):
                for j in range(a):
                    M[:, i, j] = V[:, i, j] @ U[i, j]                                  # fp32 GEMM over the channels
            t2 = np.einsum('pi,bijn->bpjn', At, M).astype(f)
            y[:, m * ty:m * ty + m, m * tx:m * tx + m, :] = np.einsum('bpjn,qj->bpqn', t2, At).astype(f)
    return y[:, :H, :W]


print("layer        |y|max   direct fp32     F(2x2,3x3) fp32   F(4x4,3x3) fp32   (max abs error vs float64; relative to |y|max in brackets)")
for (C, N, std, name) in [(768, 72, 0.002, 'convdet'), (96, 384, 0.005, 'fire13 e3'), (48, 192, 0.01, 'fire9 e3'), (64, 256, 0.03, 'fire11 e3'), (16, 64, 0.05, 'fire3 e3')]:
    x = np.maximum(rng.standard_normal((2, 12, 16, C)), 0).astype(f) * 3
    w = (rng.standard_normal((N, C, 3, 3)) * std).astype(f)
    t = direct(x.astype(np.float64), w.astype(np.float64), np.float64)
    s = np.abs(t).max()
    e = [np.abs(v - t).max() for v in (direct(x, w, f), wino(x, w, F23), wino(x, w, F43))]
    print(f"{name:10s} {s:8.3f}   " + "   ".join(f"{v:.2e} ({v / s:.1e})" for v in e))
