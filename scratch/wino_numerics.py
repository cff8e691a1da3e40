"""fp32 Winograd F(2x2,3x3) error vs direct fp32 conv, both against an fp64 truth (numpy, CPU)."""
import numpy as np
rng = np.random.default_rng(0)
def direct(x, w, dt):
    B,H,W,C = x.shape; N = w.shape[0]
    xp = np.zeros((B,H+2,W+2,C), dt); xp[:,1:-1,1:-1] = x
    y = np.zeros((B,H,W,N), dt)
    for r in range(3):
        for s in range(3):
            y += (xp[:,r:r+H,s:s+W,:].reshape(-1,C) @ w[:,:,r,s].T.astype(dt)).reshape(B,H,W,N)
    return y
def wino(x, w):
    f = np.float32
    B,H,W,C = x.shape; N = w.shape[0]
    G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], f)
    Bt = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], f)
    At = np.array([[1,1,1,0],[0,1,-1,-1]], f)
    U = np.einsum('ir,ncrs,js->ijcn', G, w.astype(f), G).astype(f)         # [4,4,C,N]
    Hp, Wp = (H+1)//2*2, (W+1)//2*2
    xp = np.zeros((B,Hp+2,Wp+2,C), f); xp[:,1:H+1,1:W+1] = x
    y = np.zeros((B,Hp,Wp,N), f)
    for ty in range(Hp//2):
        d = np.stack([xp[:,2*ty+i,:,:] for i in range(4)], 1)              # [B,4,Wp+2,C]
        for tx in range(Wp//2):
            dd = d[:,:,2*tx:2*tx+4,:]                                       # [B,4,4,C]
            V = np.einsum('ia,bakc,jk->bijc', Bt, dd, Bt).astype(f)
            M = np.einsum('bijc,ijcn->bijn', V, U).astype(f)
            Y = np.einsum('pi,bijn,qj->bpqn', At, M, At).astype(f)
            y[:,2*ty:2*ty+2,2*tx:2*tx+2,:] = Y
    return y[:,:H,:W]
for (C,N,std,name) in [(768,72,0.002,'convdet'),(96,384,0.005,'fire10 e3'),(16,64,0.05,'fire2 e3'),(64,256,0.03,'fire8')]:
    x = np.maximum(rng.standard_normal((1,12,14,C)),0).astype(np.float32)*3
    w = (rng.standard_normal((N,C,3,3))*std).astype(np.float32)
    t = direct(x.astype(np.float64), w.astype(np.float64), np.float64)
    d32 = direct(x, w, np.float32); wn = wino(x, w)
    s = np.abs(t).max()
    print(f"{name:10s} |y|max {s:8.3f}  direct err {np.abs(d32-t).max():.2e} ({np.abs(d32-t).max()/s:.1e} rel)   winograd err {np.abs(wn-t).max():.2e} ({np.abs(wn-t).max()/s:.1e} rel)")
