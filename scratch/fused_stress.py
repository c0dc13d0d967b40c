# Repeated runs of the fused Winograd kernel at the headline shapes must be bit-identical (race check).
import sys, torch
sys.path.insert(0, 'kidney-diffusion_amd')
from imagen_pytorch import _engine as E
lib = E.load(); dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for (B, H, W, Ci, Co) in [(16, 256, 256, 128, 128), (16, 128, 128, 256, 256), (16, 64, 64, 512, 512), (8, 256, 256, 256, 128)]:
    x = torch.randn(B, H, W, Ci, device=dev, generator=g)
    w = torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * (Ci * 9) ** -0.5
    b = torch.randn(Co, device=dev, generator=g)
    ys = []
    y0 = torch.empty(B, H, W, Co, device=dev)
    E.check(lib.kd_conv3x3_winograd_fused_nhwc(E.ptr(x), E.ptr(w), E.ptr(b), None, E.ptr(y0), B, H, W, Ci, Co, E.current_stream()))
    bad = 0
    y = torch.empty_like(y0)
    for it in range(25):
        y.fill_(float('nan'))
        E.check(lib.kd_conv3x3_winograd_fused_nhwc(E.ptr(x), E.ptr(w), E.ptr(b), None, E.ptr(y), B, H, W, Ci, Co, E.current_stream()))
        bad += int(not torch.equal(y, y0))
    # spot check against torch on a crop
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    err = float((y0[:1].permute(0, 3, 1, 2).double() - ref).norm() / ref.norm())
    print((B, H, W, Ci, Co), 'non-identical repeats:', bad, 'rel-L2 vs fp64 (image 0):', err)
