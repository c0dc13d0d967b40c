"""GPU parity of the individual HIP kernels against plain torch fp32 ops (the oracle's building
blocks), called through the C ABI (include/kd_engine.h)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from imagen_pytorch import _engine as E

    return E.load()


def _E():
    from imagen_pytorch import _engine as E

    return E


def g(seed):
    return torch.Generator().manual_seed(seed)


# fp32 tolerance for one contraction: |err| <= tol * sum|a||b| bound, expressed as rel-L2
CONV_REL = 2e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,stride,pad,act", [
    (2, 16, 16, 32, 64, 3, 1, 1, 0),      # plain 3x3
    (1, 8, 8, 256, 128, 3, 1, 1, 0),      # deep K
    (3, 9, 7, 36, 40, 3, 1, 1, 1),        # ragged sizes, Cin not /32, Cout not /32, SiLU
    (2, 16, 16, 64, 96, 1, 1, 0, 2),      # 1x1 + GELU
    (2, 16, 16, 32, 48, 2, 2, 0, 0),      # 2x2 stride 2 (pixel-unshuffle downsample)
    (2, 32, 32, 64, 128, 4, 2, 1, 0),     # 4x4 stride 2 pad 1: the Downsample of earlier library versions (fast path)
    (1, 10, 14, 12, 24, 4, 2, 1, 0),      # ... ragged, generic path
    (1, 20, 20, 12, 16, 7, 1, 3, 0),      # init-conv shapes: tiny Cin, wide window
    (1, 20, 20, 12, 16, 15, 1, 7, 0),
    (2, 8, 8, 132, 3, 3, 1, 1, 0),        # final conv: Cout = 3
    (1, 1, 40, 128, 512, 1, 1, 0, 0),     # token GEMM, M = 40 < tile
    (16, 16, 16, 128, 128, 3, 1, 1, 0),   # several M tiles x 1 N tile (buffer-DMA fast path, 128x128 tiles)
    (3, 17, 13, 64, 200, 3, 1, 1, 1),     # fast path, ragged M and N edges, SiLU
    (2, 32, 32, 96, 136, 2, 2, 0, 0),     # fast path, stride 2
    (5, 8, 8, 32, 72, 3, 1, 1, 0),        # fast path, tiles span several small images
    (8, 128, 128, 32, 128, 3, 1, 1, 0),   # fast path, 256x128 tiles / 8 waves
    (2, 64, 64, 160, 128, 1, 1, 0, 2),    # fast path, 1x1 + GELU
    (2, 20, 24, 12, 32, 15, 1, 7, 0x100),  # row-run K layout (init conv k=15 over 12 padded channels)
    (1, 33, 17, 12, 16, 7, 1, 3, 0x100),   # row-run, k=7, ragged image
    (3, 16, 16, 8, 64, 3, 1, 1, 0x101),    # row-run, k=3, SiLU
    (1, 8, 8, 1024, 1024, 3, 1, 1, 0),     # split-K: one batch-1 patch at the 8x8 level (M = 64, 288 K-chunks)
    (1, 9, 7, 512, 200, 3, 1, 1, 1),       # split-K, ragged M = 63 and Cout = 200, SiLU after the reduction
    (1, 16, 16, 512, 320, 1, 1, 0, 2),     # split-K of a 1x1 conv (16 chunks), GELU
    (2, 8, 8, 96, 128, 2, 2, 0, 0),        # M = 32, K too short to split: generic small-M kernel
])
def test_conv_igemm(lib, device, B, H, W, Cin, Cout, K, stride, pad, act):
    E = _E()
    x = torch.randn(B, Cin, H, W, generator=g(1))
    w = torch.randn(Cout, Cin, K, K, generator=g(2)) * (Cin * K * K) ** -0.5
    b = torch.randn(Cout, generator=g(3))
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    ref32 = F.conv2d(x, w, b, stride=stride, padding=pad)
    if act & 0xff == 1:
        ref, ref32 = F.silu(ref), F.silu(ref32)
    elif act & 0xff == 2:
        ref, ref32 = F.gelu(ref), F.gelu(ref32)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    Ho, Wo = ref.shape[-2:]
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=device)
    wd, bd = w.to(device), b.to(device)  # keep the device tensors alive across the call
    E.check(lib.kd_conv2d_nhwc(E.ptr(xd), E.ptr(wd), E.ptr(bd), E.ptr(y), B, H, W, Cin, Cout, K, K, stride, pad, act,
                               E.current_stream()))
    got = y.permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    err = float((got.double() - ref).norm() / ref.norm())
    err_cpu = float((ref32.double() - ref).norm() / ref.norm())
    # the engine must be as close to the fp64 truth as the CPU fp32 path is (x3 slack), and within CONV_REL*sqrt(K)
    assert err <= max(3 * err_cpu, CONV_REL), (err, err_cpu)


@pytest.mark.parametrize("B,S,n3,n7,n15", [
    (2, 64, 64, 32, 32),      # dim 128: the SR UNets' init conv (two 32-row tiles for the k = 3 conv)
    (1, 32, 16, 8, 8),        # dim 32 (test UNets): partial 32-row tiles
    (3, 96, 32, 16, 16),      # dim 64, S not a power of two, tiles cross image borders on every side
])
def test_init_cross_embed_conv_fused(lib, device, B, S, n3, n7, n15):
    """The three init convs (k = 3 / 7 / 15, zero padding k // 2) over a 3-plane NCHW image in one persistent
    kernel, against torch conv2d in fp64."""
    E = _E()
    x = torch.randn(B, 3, S, S, generator=g(1))
    ws = [torch.randn(n, 3, k, k, generator=g(10 + k)) * (3 * k * k) ** -0.5 for n, k in ((n3, 3), (n7, 7), (n15, 15))]
    b = torch.randn(n3 + n7 + n15, generator=g(4))
    parts64 = [F.conv2d(x.double(), w.double(), padding=w.shape[-1] // 2) for w in ws]
    ref = torch.cat(parts64, 1) + b.double()[None, :, None, None]
    ref32 = torch.cat([F.conv2d(x, w, padding=w.shape[-1] // 2) for w in ws], 1) + b[None, :, None, None]
    xd, bd = x.to(device), b.to(device)
    wd = [w.to(device) for w in ws]
    y = torch.full((B, S, S, n3 + n7 + n15), float("nan"), device=device)
    E.check(lib.kd_init_conv_nchw(E.ptr(xd), E.ptr(wd[0]), E.ptr(wd[1]), E.ptr(wd[2]), E.ptr(bd), E.ptr(y), B, S, n3, n7,
                                  n15, E.current_stream()))
    got = y.permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    err = float((got.double() - ref).norm() / ref.norm())
    err_cpu = float((ref32.double() - ref).norm() / ref.norm())
    assert err <= max(3 * err_cpu, CONV_REL), (err, err_cpu)
    # per conv: nothing leaks between the channel groups
    for sl, p64 in zip((slice(0, n3), slice(n3, n3 + n7), slice(n3 + n7, None)), parts64):
        e = float((got[:, sl].double() - (p64 + b.double()[None, sl, None, None])).norm() / p64.norm())
        assert e <= max(3 * err_cpu, CONV_REL), (sl, e)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (2, 64, 64, 128, 512),    # register epilogue: Co = 128, Wo = 64 (the SR UNet's 128 -> 256 upsample, scaled down)
    (1, 32, 96, 64, 128),     # register epilogue, Co = 32 (one lane group per (i, j)), H != W
    (2, 16, 16, 64, 192),     # Wo = 16, Co = 48: staged through LDS
    (1, 8, 8, 512, 256),      # small M: split-K, then the reduction kernel's epilogue
])
def test_conv1x1_pixel_shuffle_upsample(lib, device, B, H, W, Cin, Cout):
    """Upsample = conv1x1(C -> 4 Co) + PixelShuffle(2): the conv's epilogue writes the shuffled map directly."""
    E = _E()
    x = torch.randn(B, Cin, H, W, generator=g(1))
    w = torch.randn(Cout, Cin, 1, 1, generator=g(2)) * Cin ** -0.5
    b = torch.randn(Cout, generator=g(3))
    ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double()), 2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    wd, bd = w.to(device), b.to(device)
    y = torch.full((B, 2 * H, 2 * W, Cout // 4), float("nan"), device=device)
    E.check(lib.kd_conv2d_nhwc(E.ptr(xd), E.ptr(wd), E.ptr(bd), E.ptr(y), B, H, W, Cin, Cout, 1, 1, 1, 0, 0x200,
                               E.current_stream()))
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - ref).norm() / ref.norm()) <= CONV_REL


# Winograd F(2x2,3x3) re-associates the sum (4x4 tile transforms): fp32 error a few x the direct conv's
WINO_REL = 4e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (16, 16, 16, 512, 512),    # Mt = 1024: the SR UNet's deepest level
    (4, 32, 32, 64, 96),       # ragged N tile, several images per weight slab
    (1, 64, 32, 32, 40),       # Mt = 512, one image
    (4, 16, 16, 1024, 128),    # long K, Mt = 256 (one 256-row tile per slab)
])
def test_conv3x3_winograd_matches_direct(lib, device, B, H, W, Cin, Cout):
    E = _E()
    x = torch.randn(B, Cin, H, W, generator=g(1))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(2)) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g(3))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    wd, bd = w.to(device), b.to(device)
    y = torch.full((B, H, W, Cout), float("nan"), device=device)
    yd = torch.full((B, H, W, Cout), float("nan"), device=device)
    E.check(lib.kd_conv3x3_winograd_nhwc(E.ptr(xd), E.ptr(wd), E.ptr(bd), E.ptr(y), B, H, W, Cin, Cout,
                                         E.current_stream()))
    E.check(lib.kd_conv2d_nhwc(E.ptr(xd), E.ptr(wd), E.ptr(bd), E.ptr(yd), B, H, W, Cin, Cout, 3, 3, 1, 1, 0,
                                E.current_stream()))
    got, direct = y.permute(0, 3, 1, 2).cpu().double(), yd.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).norm() / ref.norm())
    err_direct = float((direct - ref).norm() / ref.norm())
    assert err <= WINO_REL, (err, err_direct)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "element-wise outlier"


# Winograd F(4x4,3x3): transform matrices with entries up to 8 and 1/24 - fp32 re-association error 6-8x that of
# F(2x2,3x3) (scratch/wino43_accuracy.py: 3-4e-6 at Cin = 512 .. 2048); still inside the 2e-5 of one UNet forward
WINO4_REL = 8e-6


# x3: the 36 position GEMMs on the bf16 matrix pipe (three bf16 pieces per fp32 operand, six exact products per k-step,
# fp32 accumulation - kernels_gemm_bf16x3.hip), the plan's default where tiles % 256 == 0 and Cout % 128 == 0: held to the
# SAME bounds as the fp32 MFMA GEMMs
@pytest.mark.parametrize("B,H,W,Cin,Cout,res,stats,x3", [
    (16, 16, 16, 1024, 1024, False, True, 0),   # the SR UNet's deepest level at the benchmark's batch: Mt = 256
    (16, 16, 16, 1024, 1024, False, True, 1),
    (16, 32, 32, 512, 512, True, True, 0),      # its 32x32 level, residual in the output transform
    (16, 32, 32, 512, 512, True, True, 1),
    (4, 32, 32, 64, 192, True, False, 0),       # Mt = 256 over four images, three 64-channel blocks
    (1, 64, 64, 32, 64, False, True, 0),        # one image, Mt = 256, short K
    (16, 16, 16, 2048, 128, False, True, 0),    # the concat convs' K
    (16, 16, 16, 2048, 128, False, True, 1),    # ... one 128-column tile, 128 k-stages
    (8, 16, 16, 512, 256, True, True, 0),       # Mt = 128: one 128-row tile per weight slab (batch 8 at the 16x16 level)
    (16, 16, 16, 768, 768, True, True, 0),      # configs[1] (train.py:30-39, dim_mults 1,2,3,4) at batch 16: the 16x16 up level,
    (16, 16, 16, 768, 768, True, True, 1),
    (16, 16, 16, 1280, 768, False, True, 0),    # ... its concat conv (768 + 512 skip channels), twelve 64-channel blocks
    (16, 16, 16, 1280, 768, False, True, 1),
    (16, 16, 16, 1536, 768, False, False, 0),   # K = 1536 (48 chunks of 32)
    (4, 32, 32, 32, 128, True, True, 1),        # the shortest K the bf16x3 kernel takes: two stages (its ring holds four)
    (4, 32, 32, 64, 256, False, True, 1),       # four stages
    (16, 64, 64, 512, 256, True, True, 1),      # Mt = 4096: sixteen 256-row tiles per position, 1152 tiles
    (16, 32, 32, 512, 512, True, True, 2),      # x3 = 2: V as fp32, split by the GEMM's loader waves on the way into LDS
    (16, 16, 16, 1280, 768, False, True, 2),
])
def test_conv3x3_winograd4_matches_direct(lib, device, B, H, W, Cin, Cout, res, stats, x3):
    E = _E()
    h = torch.randn(B, Cin, H, W, generator=g(1)) * 1.2 + 0.2
    x = F.silu(h)                       # what these layers see: an activated, normalised map
    w = torch.randn(Cout, Cin, 3, 3, generator=g(2)) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g(3))
    r = torch.randn(B, Cout, H, W, generator=g(4)) if res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    wd, bd = w.to(device), b.to(device)
    rd = r.permute(0, 2, 3, 1).contiguous().to(device) if res else None
    G = 8
    want_stats = stats and (Cout // G) % 16 == 0
    y = torch.full((B, H, W, Cout), float("nan"), device=device)
    ostats = torch.full((B, G, 2), float("nan"), device=device)
    call = lambda out: E.check(lib.kd_conv3x3_winograd4_nhwc(
        E.ptr(xd), E.ptr(wd), E.ptr(bd), E.ptr(rd) if res else None, E.ptr(out), B, H, W, Cin, Cout, G, 1e-5,
        E.ptr(ostats) if want_stats else None, x3, E.current_stream()))
    call(y)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).norm() / ref.norm())
    print(f"F(4x4,3x3){' bf16x3' if x3 else ''} Cin {Cin} {H}x{W}: rel-L2 {err:.2e}, max {float((got - ref).abs().max() / ref.abs().max()):.2e}")
    assert err <= WINO4_REL, err
    assert float((got - ref).abs().max()) <= 5e-5 * float(ref.abs().max()), "element-wise outlier"
    if want_stats:
        grp = ref.reshape(B, G, -1)
        got_s = ostats.cpu().double()
        assert torch.allclose(got_s[..., 0], grp.mean(dim=-1), rtol=0, atol=1e-5 * float(ref.abs().max()))
        assert torch.allclose(got_s[..., 1], (grp.var(dim=-1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-5, atol=0)
    y2 = torch.empty_like(y)
    call(y2)
    assert torch.equal(y, y2)


def test_conv3x3_winograd4_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(16, device=device)
    for shape in [(1, 18, 16, 32, 64), (1, 16, 16, 32, 64), (16, 16, 16, 48, 64), (16, 16, 16, 32, 96)]:
        rc = lib.kd_conv3x3_winograd4_nhwc(E.ptr(t), E.ptr(t), None, None, E.ptr(t), *shape, 8, 1e-5, None, 0, E.current_stream())
        assert rc != 0 and b"F(4x4,3x3)" in lib.kd_last_error()
    # shapes the fp32 path takes and the bf16x3 GEMM's tile does not: refused, not silently computed another way
    for shape in [(8, 16, 16, 512, 256), (16, 16, 16, 512, 192)]:
        rc = lib.kd_conv3x3_winograd4_nhwc(E.ptr(t), E.ptr(t), None, None, E.ptr(t), *shape, 8, 1e-5, None, 1, E.current_stream())
        assert rc != 0 and b"bf16x3" in lib.kd_last_error()


# The GEMM by itself against fp64, next to torch's fp32 product of the same operands: the six-product bf16 form is held
# to the fp32 product's error (it measures at or below it), element-wise and in the norm.
@pytest.mark.parametrize("G,M,N,K,scale_b", [
    (36, 256, 128, 32, 1.0),       # one tile per position, two k-stages
    (3, 512, 256, 96, 0.05),       # six stages (the ring wraps), 2 x 2 tiles
    (36, 1024, 512, 512, 0.05),    # the 32x32 level's GEMMs at batch 16
    (2, 256, 128, 4096, 1e-3),     # a long accumulation
    (5, 256, 384, 64, 30.0),       # a grid that is not a multiple of 8 (identity workgroup order)
    (36, 256, 1024, 1024, 0.05),   # the 16x16 level's GEMMs at batch 16: 288 tiles = one round on 256 CUs + 32 tiles cut in 8
    (36, 1024, 512, 256, 0.05),    # 576 tiles: two rounds + 64 tiles cut in 2 (sixteen k-stages: parts of eight)
    (13, 512, 384, 128, 1.0),      # 78 tiles: fewer than CUs, one tile per workgroup
    (33, 2048, 128, 64, 1.0),      # 264 tiles: one round + 8 tiles, four k-stages: too short to cut (S = 1)
])
@pytest.mark.parametrize("a_planes", [1, 0])   # 1: A in plane form (the plan's); 0: A as fp32, split by the kernel's loader waves
def test_gemm_bf16x3_matches_fp64(lib, device, G, M, N, K, scale_b, a_planes):
    E = _E()
    a = (torch.randn(G, M, K, generator=g(5)) * torch.logspace(-3, 3, K)).to(device)   # 6 decades along k
    b = (torch.randn(G, N, K, generator=g(6)) * scale_b).to(device)
    a[0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, 1e30], device=device)   # zeros, a tiny and a huge value
    c = torch.full((G, M, N), float("nan"), device=device)
    E.check(lib.kd_gemm_bf16x3(E.ptr(a), E.ptr(b), E.ptr(c), G, M, N, K, a_planes, E.current_stream()))
    ref = torch.bmm(a.double(), b.double().transpose(1, 2))
    f32 = torch.bmm(a, b.transpose(1, 2)).double()
    assert torch.isfinite(c).all()
    bound = torch.bmm(a.double().abs(), b.double().abs().transpose(1, 2))   # sum_k |a| |b|: what the errors scale with
    e_x3 = ((c.double() - ref).abs() / bound.clamp_min(1e-300))
    e_32 = ((f32 - ref).abs() / bound.clamp_min(1e-300))
    print(f"bf16x3 G{G} M{M} N{N} K{K}: max err / sum|a||b| {float(e_x3.max()):.2e} (fp32 bmm {float(e_32.max()):.2e}), "
          f"rms {float(e_x3.pow(2).mean().sqrt()):.2e} (fp32 {float(e_32.pow(2).mean().sqrt()):.2e})")
    assert float(e_x3.max()) <= max(2.0 * float(e_32.max()), 2.0 ** -23), float(e_x3.max())
    rms = lambda e: float(e.pow(2).mean().sqrt())   # (per element, relative to its own sum |a| |b|: one huge row does not decide it)
    # (at K >= 96 the six-product form measures below the fp32 product; at K = 32 - two k-steps, where the order of the six
    # terms inside a step still shows - 1.4 x it, 3.4e-8)
    assert rms(e_x3) <= max(1.5 * rms(e_32), 2.0 ** -25), (rms(e_x3), rms(e_32))   # 2^-25: half an fp32 ulp of sum |a| |b|
    c2 = torch.empty_like(c)
    E.check(lib.kd_gemm_bf16x3(E.ptr(a), E.ptr(b), E.ptr(c2), G, M, N, K, 1 - a_planes, E.current_stream()))
    assert torch.equal(c, c2)   # the other form of the A operand: the same planes, the same bits


# The same kernel in its token-GEMM / 1x1-conv form (kd_unet_config_t::x3_linear): every epilogue variant the plan uses -
# bias, residual, GlobalContext gate, strided input / output rows (skip slices) - on shapes that run whole rounds, whole
# rounds plus k-cut left-over tiles, and (fewer tiles than CUs) every tile cut in k with the epilogue in the summing launch.
@pytest.mark.parametrize("M,N,K,bias,res,gate,ldx,ldy", [
    (4096, 1024, 1024, False, True, False, 0, 0),     # feed-forward's second Linear + residual: 128 tiles, all cut in two
    (4096, 2048, 1024, False, False, False, 0, 0),    # feed-forward's first Linear (GELU applied by the next LayerNorm): one round
    (4096, 640, 1024, False, False, False, 0, 0),     # the stacked q / kv projection: 80 tiles
    (4096, 1024, 512, True, False, False, 0, 2048),   # attention's to_out into a skip slot (row stride 2048): K = 512, parts of 16 stages
    (4096, 1024, 2048, True, False, True, 2560, 0),   # the 1x1 skip conv of a ResnetBlock with the GlobalContext gate, strided input
    (16384, 512, 512, False, True, False, 0, 0),      # 32 x 32 level cross-attention projections: 256 tiles
    (768, 384, 96, True, True, False, 0, 0),          # 9 tiles, six stages: too short to cut (one tile per workgroup)
    (73728, 128, 64, True, False, False, 0, 0),       # 288 tiles: a whole round and 32 left-over tiles that are not cut
])
def test_linear_bf16x3_epilogues_match_fp64(lib, device, M, N, K, bias, res, gate, ldx, ldy):
    E = _E()
    lda, ldo = ldx or K, ldy or N
    hw = 256
    xfull = (torch.randn(M, lda, generator=g(41)) * torch.logspace(-2, 2, lda)).to(device)
    w = (torch.randn(N, K, generator=g(42)) * 0.05).to(device)
    b = torch.randn(N, generator=g(43)).to(device) if bias else None
    r = torch.randn(M, N, generator=g(44)).to(device) if res else None
    gs = torch.randn(M, N, generator=g(45)).to(device) if gate else None
    gt = torch.rand(M // hw, N, generator=g(46)).to(device) if gate else None
    y = torch.full((M, ldo), float("nan"), device=device)
    # the GroupNorm partials the launch leaves (the plan's want_seg layers): a chunk per 32 rows from the kernel's epilogue,
    # per 8 rows where the tiles are cut in k and the summing launch leaves them
    rows = lib.kd_linear_bf16x3_seg_rows(M, N, K)
    assert rows == (8 if (M // 256) * (N // 128) < 256 and K >= 256 else 32) or (M // 256) * (N // 128) % 256
    seg = torch.full((M // hw, N // 16, hw // rows, 2), float("nan"), device=device, dtype=torch.float64)
    E.check(lib.kd_linear_bf16x3(E.ptr(xfull), lda, E.ptr(w), E.ptr(b), E.ptr(r), N, E.ptr(gs), N, E.ptr(gt), hw, E.ptr(y), ldo,
                                 M, N, K, 0, 0, C.c_void_p(seg.data_ptr()), E.current_stream()))
    yy = y[:, :N].double().reshape(M // hw, hw, N // 16, 16)
    assert torch.isfinite(seg).all()
    s1, s2 = seg[..., 0].sum(-1), seg[..., 1].sum(-1)
    w1, w2 = yy.sum((1, 3)), (yy * yy).sum((1, 3))
    assert torch.allclose(s1, w1, rtol=1e-6, atol=1e-6 * float(yy.abs().sum((1, 3)).max())), float((s1 - w1).abs().max())
    assert torch.allclose(s2, w2, rtol=1e-5), float((s2 / w2 - 1).abs().max())
    x = xfull[:, :K]
    ref = x.double() @ w.double().T
    f32 = (x @ w.T).double()
    bound = x.double().abs() @ w.double().abs().T
    extra = torch.zeros_like(ref)
    if bias:
        extra = extra + b.double()
    if gate:
        extra = extra + gs.double() * gt.double().repeat_interleave(hw, dim=0)
    if res:
        extra = extra + r.double()
    got = y[:, :N].double()
    assert torch.isfinite(got).all()
    if ldo > N:
        assert torch.isnan(y[:, N:]).all()   # nothing outside the slice is written
    # the product is held to the fp32 product's error (as test_gemm_bf16x3_matches_fp64); the added terms cost at most a few
    # ulps of the sum
    scale = bound + extra.abs()
    e_x3 = ((got - (ref + extra)).abs() / scale.clamp_min(1e-300))
    e_32 = ((f32 + extra - (ref + extra)).abs() / scale.clamp_min(1e-300))
    print(f"linear bf16x3 M{M} N{N} K{K}: max err {float(e_x3.max()):.2e} (fp32 matmul {float(e_32.max()):.2e})")
    assert float(e_x3.max()) <= max(2.0 * float(e_32.max()), 2.0 ** -21), float(e_x3.max())
    rms = lambda e: float(e.pow(2).mean().sqrt())
    assert rms(e_x3) <= max(1.5 * rms(e_32), 2.0 ** -24), (rms(e_x3), rms(e_32))


def test_linear_bf16x3_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(16, device=device)
    for M, N, K, hw in [(128, 128, 64, 256), (256, 64, 64, 256), (256, 128, 48, 256)]:
        rc = lib.kd_linear_bf16x3(E.ptr(t), 0, E.ptr(t), None, None, 0, None, 0, None, hw, E.ptr(t), 0, M, N, K, 0, 0, None, E.current_stream())
        assert rc != 0 and b"kd_linear_bf16x3" in lib.kd_last_error()
    # a gate needs whole images per 256-row tile
    rc = lib.kd_linear_bf16x3(E.ptr(t), 0, E.ptr(t), None, None, 0, E.ptr(t), 128, E.ptr(t), 64, E.ptr(t), 0, 256, 128, 64, 0, 0, None,
                              E.current_stream())
    assert rc != 0
    # PixelShuffle output: maps whose width is a multiple of 16, channel runs of 32 per sub-position
    for M, N, K, wo in [(16384, 64, 64, 64), (16384, 256, 64, 40)]:
        rc = lib.kd_linear_bf16x3(E.ptr(t), 0, E.ptr(t), None, None, 0, None, 0, None, 256, E.ptr(t), 0, M, N, K, 1, wo, None,
                                  E.current_stream())
        assert rc != 0


@pytest.mark.parametrize("B,H,W,K,Co,ldy,act", [
    (16, 64, 64, 256, 128, 256, 1),    # the 64 x 64 level's upsample of the SR UNet into the first half of a concat buffer
    (4, 32, 32, 1024, 512, 0, 1),      # the 32 x 32 level's shape at batch 4: 256 tiles
    (16, 16, 16, 1024, 512, 0, 1),     # the 16 x 16 level's: an accumulator block's 32 rows are two image rows
    (4, 64, 32, 64, 32, 0, 2),         # a narrow one (Co = 32: one channel run per sub-position), GELU
])
def test_linear_bf16x3_upsample_form_matches_fp64(lib, device, B, H, W, K, Co, ldy, act):
    """conv1x1 -> SiLU -> PixelShuffle(2) of the library's Upsample through the bf16x3 kernel's epilogue: weight
    rows packed n' = (2 i + j) Co + c as the plan packs them, output [B, 2H, 2W, Co] written with row stride ldy."""
    E = _E()
    M, N = B * H * W, 4 * Co
    x = torch.randn(M, K, generator=g(51)).to(device)
    wt = (torch.randn(N, K, generator=g(52)) * 0.1).to(device)        # torch layout: row c * 4 + i * 2 + j
    bt = torch.randn(N, generator=g(53)).to(device)
    perm = torch.tensor([(n % Co) * 4 + n // Co for n in range(N)], device=device)   # packed row n' = q Co + c <- torch row c 4 + q
    wp, bp = wt[perm].contiguous(), bt[perm].contiguous()
    ldo = ldy or Co
    y = torch.full((4 * M, ldo), float("nan"), device=device)
    seg = torch.full((B, Co // 16, (H * W // 32) * 4, 2), float("nan"), device=device, dtype=torch.float64)
    E.check(lib.kd_linear_bf16x3(E.ptr(x), 0, E.ptr(wp), E.ptr(bp), None, 0, None, 0, None, H * W, E.ptr(y), ldo, M, N, K, act, W,
                                 C.c_void_p(seg.data_ptr()), E.current_stream()))
    z = x.double() @ wt.double().T + bt.double()
    z = F.silu(z) if act == 1 else F.gelu(z)
    ref = F.pixel_shuffle(z.reshape(B, H, W, N).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)   # [B, 2H, 2W, Co]
    got = y[:, :Co].double().reshape(B, 2 * H, 2 * W, Co)
    assert torch.isfinite(got).all()
    if ldo > Co:
        assert torch.isnan(y[:, Co:]).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    f32 = x @ wt.T + bt
    f32 = (F.silu(f32) if act == 1 else F.gelu(f32)).double()
    e32 = float((f32 - z).abs().max() / z.abs().max())
    print(f"upsample form M{M} K{K} Co{Co}: max err / max |y| {err:.2e} (fp32 torch {e32:.2e})")
    assert err <= max(3.0 * e32, 2e-6), (err, e32)
    assert torch.isfinite(seg).all()
    s1, s2 = seg[..., 0].sum(-1), seg[..., 1].sum(-1)
    gg = got.reshape(B, 4 * H * W, Co // 16, 16)
    assert torch.allclose(s1, gg.sum((1, 3)), rtol=1e-6, atol=1e-6 * float(gg.abs().sum((1, 3)).max()))
    assert torch.allclose(s2, (gg * gg).sum((1, 3)), rtol=1e-5)


@pytest.mark.parametrize("B,H,W,Cin,O,ldx", [
    (16, 64, 64, 128, 256, 0),     # the SR UNet's 128 x 128 -> 64 x 64 downsample at a quarter of the size: K = 512, 128 tiles cut in k
    (4, 128, 64, 256, 128, 384),   # K = 1024, input a channel slice of a wider buffer (row stride 384), 32 x ... 8192 rows: 32 tiles
    (16, 32, 32, 64, 384, 0),      # K = 256
])
def test_downsample_bf16x3_matches_fp64(lib, device, B, H, W, Cin, O, ldx):
    """Downsample of the library (pixel-unshuffle + conv1x1) through the bf16x3 kernel whose loader gathers the 2 x 2 input
    pixels: against the same op in fp64 from the torch weight layout [O][4 C] (k = c 4 + 2 s1 + s2)."""
    E = _E()
    ld = ldx or Cin
    xf = torch.randn(B, H, W, ld, generator=g(61)).to(device)
    wt = (torch.randn(O, 4 * Cin, generator=g(62)) * 0.05).to(device)
    bt = torch.randn(O, generator=g(63)).to(device)
    M, hw = B * (H // 2) * (W // 2), (H // 2) * (W // 2)
    if M % 256 or (M // 256) * (O // 128) < 1:
        pytest.skip("shape outside the kernel's tiles")
    y = torch.full((M, O), float("nan"), device=device)
    rows = lib.kd_linear_bf16x3_seg_rows(M, O, 4 * Cin)
    seg = torch.full((B, O // 16, hw // rows, 2), float("nan"), device=device, dtype=torch.float64)
    E.check(lib.kd_downsample_bf16x3(E.ptr(xf), ld, E.ptr(wt), E.ptr(bt), E.ptr(y), B, H, W, Cin, O, C.c_void_p(seg.data_ptr()),
                                     E.current_stream()))
    x = xf[..., :Cin].double().permute(0, 3, 1, 2)                       # NCHW
    xu = F.pixel_unshuffle(x, 2)                                          # channel c 4 + 2 s1 + s2
    ref = F.conv2d(xu, wt.double().reshape(O, 4 * Cin, 1, 1), bt.double()).permute(0, 2, 3, 1).reshape(M, O)
    f32 = F.conv2d(xu.float(), wt.reshape(O, 4 * Cin, 1, 1), bt).permute(0, 2, 3, 1).reshape(M, O).double()
    got = y.double()
    assert torch.isfinite(got).all()
    e_x3, e_32 = float((got - ref).abs().max() / ref.abs().max()), float((f32 - ref).abs().max() / ref.abs().max())
    print(f"downsample bf16x3 B{B} {H}x{W} C{Cin}->{O}: max err / max |y| {e_x3:.2e} (fp32 torch {e_32:.2e})")
    assert e_x3 <= max(2.0 * e_32, 1e-6), (e_x3, e_32)
    yy = got.reshape(B, hw, O // 16, 16)
    s1, s2 = seg[..., 0].sum(-1), seg[..., 1].sum(-1)
    assert torch.allclose(s1, yy.sum((1, 3)), rtol=1e-6, atol=1e-6 * float(yy.abs().sum((1, 3)).max()))
    assert torch.allclose(s2, (yy * yy).sum((1, 3)), rtol=1e-5)


def test_gemm_bf16x3_two_launches_sharing_the_chip_do_not_depend_on_each_other(lib, device):
    """No workgroup of the persistent kernel waits for another (the k-parts of a left-over tile leave their accumulators in
    slabs of the launch's own workspace, a second launch adds them), so two such launches on different streams (two plans of
    a grid run) may share the CUs in any interleaving: same bits as alone."""
    import threading

    E = _E()
    shapes = [(36, 256, 1024, 1024), (36, 1024, 512, 256)]   # 288 tiles (32 cut in 8) and 576 tiles (64 cut in 2)
    data, alone = [], []
    for i, (G, M, N, K) in enumerate(shapes):
        a = torch.randn(G, M, K, generator=g(20 + i)).to(device)
        b = (torch.randn(G, N, K, generator=g(30 + i)) * 0.05).to(device)
        c = torch.empty(G, M, N, device=device)
        E.check(lib.kd_gemm_bf16x3(E.ptr(a), E.ptr(b), E.ptr(c), G, M, N, K, 1, E.current_stream()))
        data.append((a, b))
        alone.append(c.clone())
    streams = [torch.cuda.Stream(device=device) for _ in shapes]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        G, M, N, K = shapes[i]
        a, b = data[i]
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(12):
                    c = torch.full((G, M, N), float("nan"), device=device)
                    E.check(lib.kd_gemm_bf16x3(E.ptr(a), E.ptr(b), E.ptr(c), G, M, N, K, 1, streams[i].cuda_stream))
                    if not torch.equal(c, alone[i]):
                        errors.append((i, float((c - alone[i]).abs().max())))
        except Exception as e:   # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(shapes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:4]


def test_gemm_bf16x3_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(16, device=device)
    for shape in [(1, 128, 128, 32), (1, 256, 64, 32), (1, 256, 128, 16), (36, 4096 * 8, 128, 2048)]:
        rc = lib.kd_gemm_bf16x3(E.ptr(t), E.ptr(t), E.ptr(t), *shape, 0, E.current_stream())
        assert rc != 0 and b"bf16x3" in lib.kd_last_error()


def test_conv3x3_winograd_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(16, device=device)
    rc = lib.kd_conv3x3_winograd_nhwc(E.ptr(t), E.ptr(t), None, E.ptr(t), 1, 15, 16, 32, 64, E.current_stream())
    assert rc != 0 and b"even" in lib.kd_last_error()


@pytest.mark.parametrize("B,H,W,Cin,Cout,G,film,res,ldx", [
    (2, 64, 64, 128, 128, 8, True, False, 0),    # FiLM, 32 chunks (a multiple of the 4 pipeline stages)
    (1, 8, 16, 128, 128, 8, False, True, 0),     # ONE item: every side is padding (must stay 0 after the activation)
    (3, 32, 48, 40, 256, 2, True, True, 0),      # 10 chunks (padded to 12), 2 channel slabs, H != W
    (1, 48, 16, 8, 128, 2, False, False, 0),     # 2 chunks (shorter than the pipeline's prefetch)
    (2, 32, 32, 512, 128, 8, True, False, 0),    # 128 chunks, one 128-channel slab
    (6, 64, 64, 64, 256, 8, True, True, 0),      # 768 items on 256 persistent workgroups (further items per workgroup:
                                                 # prefetch under the epilogue, table written behind the exchange)
    (1, 16, 32, 68, 128, 1, False, True, 0),     # 17 chunks: one steady trip of 12 + a guarded remainder of 5; every patch a border
    (1, 32, 32, 1056, 128, 8, True, False, 0),   # affine table beyond 1024 channels (second half), 264 chunks
    (1, 16, 16, 8, 128, 2, False, False, 0),     # two items, two chunks
    (2, 16, 48, 12, 256, 1, True, True, 0),      # three chunks, two slabs; W = 3 patches
    (2, 32, 32, 128, 128, 8, True, True, 384),   # STRIDED input: a 128-channel slice at channel offset 128 of 384-float rows
                                                 # (a skip tensor living in its concat buffer)
    (1, 24, 32, 96, 128, 8, False, False, 224),  # strided input, row stride not a multiple of the slice; H = 3 patches
    (1, 16, 32, 2048, 128, 8, True, False, 0),   # Cin = 2048: the whole affine table in LDS (the 16x16 level's first conv), 512 chunks
    (2, 16, 16, 2048, 256, 8, False, True, 3072),  # Cin = 2048 read through a 3072-float row stride
])
def test_gn_conv3x3_winograd_fused_matches_torch(lib, device, B, H, W, Cin, Cout, G, film, res, ldx):
    """ResnetBlock `Block` = conv3x3(SiLU(FiLM(GroupNorm(x)))) with the activation applied to the raw patch in
    LDS inside the fused Winograd kernel (hardware exp2 / reciprocal: ~1 ulp each)."""
    E = _E()
    x = torch.randn(B, Cin, H, W, generator=g(1)) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cin, generator=g(5))
    beta = 0.2 * torch.randn(Cin, generator=g(6))
    ss = 0.3 * torch.randn(B, 2 * Cin, generator=g(7)) if film else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g(2)) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=g(3))
    r = torch.randn(B, Cout, H, W, generator=g(4)) if res else None
    h = F.group_norm(x.double(), G, gamma.double(), beta.double(), eps=1e-5)
    if film:
        h = h * (ss[:, :Cin, None, None].double() + 1) + ss[:, Cin:, None, None].double()
    ref = F.conv2d(F.silu(h), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    if ldx:   # the kernel's input is channels [c0, c0 + Cin) of rows of ldx floats; the other channels hold junk
        c0 = min(128, (ldx - Cin) // 4 * 4)
        wide = torch.full((B, H, W, ldx), 1e30, device=device)
        wide[..., c0:c0 + Cin] = xd
        xd = wide[..., c0:]          # data_ptr at channel c0 of row 0; the tensor itself is only a pointer carrier
        assert xd.data_ptr() % 16 == 0
    xptr = C.c_void_p(xd.data_ptr())
    gd, bed, wd, bd = gamma.to(device), beta.to(device), w.to(device), b.to(device)
    ssd = ss.to(device) if film else None
    rd = r.permute(0, 2, 3, 1).contiguous().to(device) if res else None
    y = torch.full((B, H, W, Cout), float("nan"), device=device)
    # the epilogue also leaves GroupNorm statistics of y (for the next layer) where its groups are 16-channel multiples
    Go = 8 if (Cout // 8) % 16 == 0 else 0
    ostats = torch.full((B, max(Go, 1), 2), float("nan"), device=device)
    call = lambda out: E.check(lib.kd_gn_conv3x3_winograd_fused_nhwc(
        xptr, E.ptr(gd), E.ptr(bed), E.ptr(ssd) if film else None, E.ptr(wd), E.ptr(bd),
        E.ptr(rd) if res else None, E.ptr(out), B, H, W, Cin, Cout, G, 1e-5, E.ptr(ostats) if Go and G == 8 else None,
        ldx, E.current_stream()))
    call(y)
    if Go and G == 8:
        grp = ref.reshape(B, Go, -1)
        want_mean = grp.mean(dim=-1)
        want_rstd = (grp.var(dim=-1, unbiased=False) + 1e-5).rsqrt()
        got_s = ostats.cpu().double()
        assert torch.allclose(got_s[..., 0], want_mean, rtol=0, atol=2e-6 * float(ref.abs().max()))
        assert torch.allclose(got_s[..., 1], want_rstd, rtol=1e-5, atol=0)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).norm() / ref.norm())
    assert err <= WINO_REL, err
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "element-wise outlier"
    y2 = torch.empty_like(y)
    call(y2)
    assert torch.equal(y, y2)


def test_gn_conv3x3_winograd_fused_at_benchmark_size_is_repeatable(lib, device):
    """The SR UNet's top level at the benchmark's batch (16 x 256 x 256 x 128 -> 128: 8192 items of 16 x 8 pixels on
    256 persistent workgroups, 32 items each): repeated launches are bit-identical (no race between the DMA stages, the
    V stores, the exchange and the barriers shows up under full occupancy) and image 0 matches an fp64 reference."""
    E = _E()
    B, H, W, Cin, Cout, G = 16, 256, 256, 128, 128, 8
    gd = torch.Generator(device=device).manual_seed(5)
    x = torch.randn(B, H, W, Cin, device=device, generator=gd)
    w = torch.randn(Cout, Cin, 3, 3, device=device, generator=gd) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, device=device, generator=gd)
    gamma = 1 + 0.2 * torch.randn(Cin, device=device, generator=gd)
    beta = 0.2 * torch.randn(Cin, device=device, generator=gd)
    call = lambda out: E.check(lib.kd_gn_conv3x3_winograd_fused_nhwc(
        E.ptr(x), E.ptr(gamma), E.ptr(beta), None, E.ptr(w), E.ptr(b), None, E.ptr(out), B, H, W, Cin, Cout, G, 1e-5, None,
        0, E.current_stream()))
    y0 = torch.empty(B, H, W, Cout, device=device)
    call(y0)
    y = torch.empty_like(y0)
    for _ in range(8):
        y.fill_(float("nan"))
        call(y)
        assert torch.equal(y, y0)
    x0 = x[:1].permute(0, 3, 1, 2).cpu().double()
    h = F.silu(F.group_norm(x0, G, gamma.cpu().double(), beta.cpu().double(), eps=1e-5))
    ref = F.conv2d(h, w.cpu().double(), b.cpu().double(), padding=1)
    got = y0[:1].permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).norm() / ref.norm()) <= WINO_REL


def test_gn_conv3x3_winograd_fused_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(4096, device=device)
    p = E.ptr(t)
    for shape, ldx, word in [((1, 20, 16, 32, 128), 0, b"% 8"), ((1, 16, 24, 32, 128), 0, b"% 16"), ((1, 16, 16, 30, 128), 0, b"% 4"),
                             ((1, 16, 16, 32, 64), 0, b"% 128"), ((1, 16, 16, 32, 192), 0, b"% 128"),
                             ((1, 16, 16, 32, 128), 16, b"row stride"), ((1, 16, 16, 32, 128), 34, b"row stride")]:
        rc = lib.kd_gn_conv3x3_winograd_fused_nhwc(p, p, p, None, p, p, None, p, *shape, 8, 1e-5, None, ldx, E.current_stream())
        assert rc != 0 and word in lib.kd_last_error(), (shape, ldx, lib.kd_last_error())


@pytest.mark.parametrize("B,HW,C,G,film", [(2, 64, 32, 8, False), (3, 100, 96, 8, True), (1, 4096, 128, 8, True),
                                           (2, 16, 1024, 8, True), (2, 300, 384, 8, False)])
def test_groupnorm_film_silu(lib, device, B, HW, C, G, film):
    E = _E()
    x = torch.randn(B, HW, C, generator=g(4)) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g(5)), torch.randn(C, generator=g(6))
    ss = torch.randn(B, 2 * C, generator=g(7)) * 0.3 if film else None
    xn = x.permute(0, 2, 1)  # [B,C,HW]
    ref = F.group_norm(xn.double(), G, gamma.double(), beta.double(), eps=1e-5)
    if film:
        sc, sh = ss.double()[:, :C, None], ss.double()[:, C:, None]
        ref = ref * (sc + 1) + sh
    ref = F.silu(ref).permute(0, 2, 1)
    y = torch.empty(B, HW, C, device=device)
    xd, gd, bd = x.to(device), gamma.to(device), beta.to(device)
    sd = ss.to(device) if film else None
    E.check(lib.kd_groupnorm_silu_nhwc(E.ptr(xd), E.ptr(gd), E.ptr(bd), E.ptr(sd), E.ptr(y), B, HW, C, G, 1e-5,
                                       E.current_stream()))
    assert torch.allclose(y.cpu().double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("rows,C,bias", [(7, 32, False), (300, 256, True), (64, 2048, False), (5, 4096, True)])
def test_layernorm(lib, device, rows, C, bias):
    E = _E()
    x = torch.randn(rows, C, generator=g(8)) * 3 + 1
    gg = torch.randn(C, generator=g(9))
    bb = torch.randn(C, generator=g(10)) if bias else None
    ref = F.layer_norm(x.double(), (C,), gg.double(), bb.double() if bias else None, eps=1e-5)
    y = torch.empty(rows, C, device=device)
    xd, gd = x.to(device), gg.to(device)
    bd = bb.to(device) if bias else None
    E.check(lib.kd_layernorm(E.ptr(xd), E.ptr(gd), E.ptr(bd), E.ptr(y), rows, C, 1e-5, E.current_stream()))
    assert torch.allclose(y.cpu().double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("rows,C,act,res,second", [
    (300, 1024, 0, True, True),     # attention's to_out LayerNorm + residual, then the feed-forward's first LayerNorm
    (64, 2048, 2, False, False),    # FeedForward: GELU of the raw product on the way in
    (5, 4096, 2, True, True), (7, 32, 1, False, True), (300, 256, 0, False, False), (33, 520, 2, True, False),
])
def test_layernorm_forms_of_the_transformer_block(lib, device, rows, C, act, res, second):
    E = _E()
    x = torch.randn(rows, C, generator=g(8)) * 3 + 1
    gg, g2 = torch.randn(C, generator=g(9)), torch.randn(C, generator=g(14))
    r = torch.randn(rows, C, generator=g(15)) if res else None
    xa = x.double()
    if act == 1:
        xa = F.silu(xa)
    elif act == 2:
        xa = F.gelu(xa)
    ref = F.layer_norm(xa, (C,), gg.double(), None, eps=1e-5)
    if res:
        ref = ref + r.double()
    y = torch.empty(rows, C, device=device)
    y2 = torch.empty(rows, C, device=device) if second else None
    xd, gd, g2d = x.to(device), gg.to(device), g2.to(device)   # (kept alive: a temporary's block would be handed to the next one)
    rd = r.to(device) if res else None
    E.check(lib.kd_layernorm_ex(E.ptr(xd), E.ptr(gd), None, E.ptr(rd), E.ptr(y), rows, C, 1e-5, act, E.ptr(g2d) if second else None,
                                E.ptr(y2), E.current_stream()))
    assert torch.allclose(y.cpu().double(), ref, rtol=2e-5, atol=2e-5)
    if second:
        ref2 = F.layer_norm(ref, (C,), g2.double(), None, eps=1e-5)
        assert torch.allclose(y2.cpu().double(), ref2, rtol=4e-5, atol=4e-5)
    # the plain form is what kd_layernorm computes, bit for bit (one read of the row instead of three: same sums, same order)
    if act == 0 and not res:
        y0 = torch.empty(rows, C, device=device)
        E.check(lib.kd_layernorm(E.ptr(xd), E.ptr(gd), None, E.ptr(y0), rows, C, 1e-5, E.current_stream()))
        assert torch.equal(y0, y)


@pytest.mark.parametrize("rows,C,N,act,res", [(4096, 512, 1664, 0, False), (4096, 512, 1024, 0, False), (4096, 1024, 512, 2, True),
                                               (16384, 256, 768, 0, True), (4096, 2048, 1024, 2, False)])
def test_layernorm_planes_feed_the_bf16x3_linear(lib, device, rows, C, N, act, res):
    """The plan's LayerNorm -> Linear pairs (engine.hip transformer): the LayerNorm leaves the GEMM's A operand as three bf16
    planes.  The planes of a value add up to the fp32 LayerNorm output exactly, and the GEMM over them is bit-identical to
    the GEMM over the fp32 rows (same products, same order; only the loader differs)."""
    E = _E()
    x = torch.randn(rows, C, generator=g(21)) * 2 + 0.5
    gg = torch.randn(C, generator=g(22))
    w = torch.randn(N, C, generator=g(23)) / C ** 0.5
    b = torch.randn(N, generator=g(24))
    r = torch.randn(rows, N, generator=g(25)) if res else None
    xd, gd, wd, bd = x.to(device), gg.to(device), w.to(device), b.to(device)
    rd = r.to(device) if res else None
    ln = torch.empty(rows, C, device=device)
    E.check(lib.kd_layernorm_ex(E.ptr(xd), E.ptr(gd), None, None, E.ptr(ln), rows, C, 1e-5, act, None, None, E.current_stream()))
    y_rows = torch.empty(rows, N, device=device)
    E.check(lib.kd_linear_bf16x3(E.ptr(ln), C, E.ptr(wd), E.ptr(bd), E.ptr(rd), N if res else 0, None, 0, None, rows, E.ptr(y_rows),
                                 N, rows, N, C, 0, 0, None, E.current_stream()))
    planes = torch.zeros(3, C // 16, rows, 16, dtype=torch.bfloat16, device=device)
    y = torch.empty(rows, N, device=device)
    E.check(lib.kd_layernorm_linear_bf16x3(E.ptr(xd), E.ptr(gd), None, rows, C, 1e-5, act, E.ptr(wd), E.ptr(bd), E.ptr(rd),
                                           N if res else 0, E.ptr(y), N, N, planes.data_ptr(), E.current_stream()))
    back = planes.double().sum(0).permute(1, 0, 2).reshape(rows, C)
    assert torch.equal(back, ln.double())
    assert torch.equal(y, y_rows)
    xa = x.double()
    if act == 2:
        xa = F.gelu(xa)
    ref = F.layer_norm(xa, (C,), gg.double(), None, eps=1e-5) @ w.double().t() + b.double()
    if res:
        ref = ref + r.double()
    assert (y.cpu().double() - ref).norm() / ref.norm() < 3e-6


@pytest.mark.parametrize("B,Nq,Nk,H,Hkv", [
    (2, 64, 69, 8, 1), (1, 300, 5, 8, 8), (2, 256, 261, 8, 1), (1, 1000, 1029, 4, 1),   # small grids: 4 lanes per query
    (1, 70, 3, 8, 1),                      # fewer keys than key splits (one split sees no key at all)
    # >= 128 blocks of 128 queries: both contractions on the matrix cores (attention_mfma_kernel)
    (16, 512, 517, 8, 1),                  # multi-query, 5 keys in the last tile
    (33, 256, 261, 8, 8),                  # per-head K/V
    (16, 500, 517, 8, 1),                  # ragged query count (last wave half empty)
    (8, 1024, 37, 8, 1),                   # fewer keys than one tile (cross-attention sized)
    (4, 512, 128, 8, 1),                   # exactly two full tiles, exactly 128 blocks
])
def test_attention(lib, device, B, Nq, Nk, H, Hkv):
    E = _E()
    D = 64
    q = torch.randn(B, Nq, H, D, generator=g(11)) * 0.5
    k = torch.randn(B, Nk, Hkv, D, generator=g(12))
    v = torch.randn(B, Nk, Hkv, D, generator=g(13))
    kk = k.expand(B, Nk, H, D) if Hkv == 1 else k
    vv = v.expand(B, Nk, H, D) if Hkv == 1 else v
    sim = torch.einsum("bihd,bjhd->bhij", q.double(), kk.double())
    ref = torch.einsum("bhij,bjhd->bihd", sim.softmax(dim=-1), vv.double())
    out = torch.empty(B, Nq, H, D, device=device)
    qd, kd, vd = q.to(device), k.to(device).contiguous(), v.to(device).contiguous()
    E.check(lib.kd_attention(E.ptr(qd), E.ptr(kd), E.ptr(vd), E.ptr(out), B, Nq, Nk, H, Hkv, D, E.current_stream()))
    assert torch.allclose(out.cpu().double(), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("B,n,q", [(3, 12288, 0.95), (2, 196608, 0.95), (1, 1000, 0.5), (2, 37, 0.0), (2, 37, 1.0),
                                   (4, 4096, 0.95)])
def test_quantile_abs_matches_torch(lib, device, B, n, q):
    E = _E()
    x = torch.randn(B, n, generator=g(14)) * 1.7
    if n == 4096:  # heavy duplicates: forces the equal-bin branch of the radix select
        x = (x * 2).round() / 2
    ref = torch.quantile(x.abs(), q, dim=-1)
    ws_bytes = lib.kd_quantile_workspace_bytes(B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    out = torch.empty(B, device=device)
    xd = x.to(device)
    E.check(lib.kd_quantile_abs(E.ptr(xd), E.ptr(out), B, n, q, C.c_void_p(ws.data_ptr()), ws_bytes,
                                E.current_stream()))
    # order statistics are selected exactly; the final lerp is one fp32 fma apart at most
    assert torch.allclose(out.cpu(), ref, rtol=0, atol=2e-7 * float(ref.abs().max()) + 1e-12), (out.cpu(), ref)


def test_philox_normal_matches_host_reference_and_is_normal(lib, device):
    from oracle.philox_ref import philox_normal

    E = _E()
    n = 1 << 20
    out = torch.empty(n, device=device)
    E.check(lib.kd_philox_normal(E.ptr(out), n, 0x1234567, (16 << 32) | 2, E.current_stream()))
    got = out.cpu().numpy()
    ref = philox_normal(n, 0x1234567, (16 << 32) | 2)
    assert np.abs(got - ref).max() < 2e-5   # same counters; only libm vs ocml ulps differ
    assert abs(got.mean()) < 5e-3 and abs(got.std() - 1) < 5e-3
    out2 = torch.empty(n, device=device)
    E.check(lib.kd_philox_normal(E.ptr(out2), n, 0x1234567, (16 << 32) | 3, E.current_stream()))
    assert abs(float((out * out2).mean())) < 5e-3  # streams are independent
