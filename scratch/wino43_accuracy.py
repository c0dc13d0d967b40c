"""fp32 accuracy of Winograd F(4x4,3x3) against F(2x2,3x3) and a direct fp32 convolution, all against an fp64 direct
convolution: decides whether the deep (Cin >= 512) ResnetBlock convs may use it under the forward tolerance (2e-5)."""
import numpy as np, torch, torch.nn.functional as F, sys
torch.manual_seed(0)
BT4 = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=np.float64)
G4 = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype=np.float64)
AT4 = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=np.float64)
BT2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=np.float64)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=np.float64)
AT2 = np.array([[1,1,1,0],[0,1,-1,-1]], dtype=np.float64)

def wino(x, w, BT, G, AT, m):
    """x [C,H,W] fp32, w [N,C,3,3] fp32 -> [N,H,W] with every step in fp32 (transforms as exact fp32 matrices)."""
    a = BT.shape[0]
    C, H, W = x.shape
    xp = np.zeros((C, H + 2 + m, W + 2 + m), np.float32); xp[:, 1:H+1, 1:W+1] = x
    BTf, Gf, ATf = BT.astype(np.float32), G.astype(np.float32), AT.astype(np.float32)
    U = np.einsum('ij,ncjk,lk->ilnc', Gf, w, Gf).astype(np.float32)          # [a,a,N,C]
    th, tw = H // m, W // m
    d = np.stack([np.stack([xp[:, i*m:i*m+a, j*m:j*m+a] for j in range(tw)]) for i in range(th)])   # [th,tw,C,a,a]
    V = np.einsum('ij,twcjk,lk->iltwc', BTf, d, BTf).astype(np.float32)       # [a,a,th,tw,C]
    M = np.zeros((a, a, th * tw, w.shape[0]), np.float32)
    for i in range(a):
        for j in range(a):
            M[i, j] = V[i, j].reshape(th * tw, C) @ U[i, j].T               # fp32 GEMM
    Y = np.einsum('ij,jktn,lk->iltn', ATf, M, ATf).astype(np.float32)         # [m,m,T,N]
    out = np.zeros((w.shape[0], H, W), np.float32)
    for t in range(th * tw):
        i, j = divmod(t, tw)
        out[:, i*m:(i+1)*m, j*m:(j+1)*m] = Y[:, :, t, :].transpose(2, 0, 1)
    return out

for C, H in ((512, 32), (1024, 16), (2048, 16)):
    N = 64
    h = torch.randn(C, H, H) * 1.2 + 0.2
    x = F.silu(h).numpy().astype(np.float32)          # what the conv sees: SiLU of a normalised map
    w = (torch.randn(N, C, 3, 3) * (9 * C) ** -0.5).numpy().astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), padding=1)[0].numpy()
    direct = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    r = lambda y: (np.linalg.norm(y.astype(np.float64) - ref) / np.linalg.norm(ref), np.abs(y - ref).max() / np.abs(ref).max())
    print(f"Cin {C} {H}x{H}: direct fp32 rel-L2 {r(direct)[0]:.2e} max {r(direct)[1]:.2e} | F(2,3) {r(wino(x, w, BT2, G2, AT2, 2))[0]:.2e} "
          f"max {r(wino(x, w, BT2, G2, AT2, 2))[1]:.2e} | F(4,3) {r(wino(x, w, BT4, G4, AT4, 4))[0]:.2e} max {r(wino(x, w, BT4, G4, AT4, 4))[1]:.2e}")
