"""Same-box A/B of the two fused ResnetBlock conv kernels (F(2x2,3x3) 128-channel items vs F(4x4,3x3)) on the SR UNet's
layer shapes at batch 16, through the C ABI; run under `rocprofv3 --kernel-trace` and read the kernel durations with
scratch/w4f_ab_parse.py (the wrappers also pack weights and compute statistics: only the conv kernels are compared)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "kidney-diffusion_amd"))
from imagen_pytorch import _engine as E  # noqa: E402

lib = E.load()
dev = torch.device("cuda:0")
SHAPES = [(16, 256, 256, 128, 128), (16, 256, 256, 256, 128), (16, 128, 128, 128, 128), (16, 128, 128, 256, 128),
          (16, 64, 64, 256, 256)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
g = torch.Generator(device=dev).manual_seed(0)
for (B, H, W, Cin, Cout) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, device=dev, generator=g)
    gamma = 1 + 0.2 * torch.randn(Cin, device=dev, generator=g)
    beta = 0.2 * torch.randn(Cin, device=dev, generator=g)
    r = torch.randn(B, H, W, Cout, device=dev, generator=g)
    ya = torch.empty(B, H, W, Cout, device=dev)
    yb = torch.empty(B, H, W, Cout, device=dev)
    st = torch.empty(B, 8, 2, device=dev)
    for it in range(3):
        for fn, y in ((lib.kd_gn_conv3x3_winograd_fused_nhwc, ya), (lib.kd_gn_conv3x3_winograd4_fused_nhwc, yb)):
            E.check(fn(E.ptr(x), E.ptr(gamma), E.ptr(beta), None, E.ptr(w), E.ptr(b), E.ptr(r), E.ptr(y), B, H, W, Cin, Cout, 8,
                       1e-5, E.ptr(st), 0, E.current_stream()))
    torch.cuda.synchronize()
    d = (ya - yb).double()
    print(f"{B}x{H}x{W} {Cin}->{Cout}: rel diff between the two kernels {float(d.norm() / ya.double().norm()):.2e}", flush=True)
    del x, w, r, ya, yb
