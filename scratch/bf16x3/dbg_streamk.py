"""Debug aid: per-tile error map of the product's stream-K bf16x3 GEMM (kd_gemm_bf16x3) against torch.bmm."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "kidney-diffusion_amd"))
import torch
from imagen_pytorch import _engine as E

lib = E.load()
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(36, 256, 1024, 1024)]
for G, M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(G, M, K, device=dev, generator=g)
    b = torch.randn(G, N, K, device=dev, generator=g) * 0.05
    c = torch.full((G, M, N), float("nan"), device=dev)
    E.check(lib.kd_gemm_bf16x3(E.ptr(a), E.ptr(b), E.ptr(c), G, M, N, K, E.current_stream()))
    ref = torch.bmm(a, b.transpose(1, 2))
    err = (c - ref).abs().reshape(G, M // 256, 256, N // 128, 128).amax(dim=(2, 4))   # [G][mt][nt]
    nanmap = torch.isnan(c).reshape(G, M // 256, 256, N // 128, 128).any(dim=2).any(dim=-1)
    bad = (err > 1e-3) | nanmap
    print(f"G{G} M{M} N{N} K{K}: {int(bad.sum())} bad tiles of {bad.numel()}, nan tiles {int(nanmap.sum())}, max err {float(err.nan_to_num(9e9).max()):.3e}")
    idx = bad.reshape(-1).nonzero().flatten().tolist()
    print("  bad tile indices (g, mt, nt order):", idx[:40])
    nk = K // 16
    tiles = bad.numel()
    P = min(tiles, 256)
    cuts = sorted({(q * tiles * nk // P) // nk for q in range(1, P) if (q * tiles * nk // P) % nk})
    print("  split tiles:", cuts[:40], "...", len(cuts))
    if idx:
        t = idx[0]
        gg, mt, nt = t // ((M // 256) * (N // 128)), (t // (N // 128)) % (M // 256), t % (N // 128)
        blk = (c - ref)[gg, mt * 256:(mt + 1) * 256, nt * 128:(nt + 1) * 128]
        rat = (c / ref)[gg, mt * 256:(mt + 1) * 256, nt * 128:(nt + 1) * 128]
        print("  first bad tile: |diff| by 64x64 wave block:", blk.abs().reshape(4, 64, 2, 64).amax(dim=(1, 3)).tolist())
        print("  median ratio c/ref:", float(rat.median()))
