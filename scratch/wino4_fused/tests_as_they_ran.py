"""The kernel tests of the fused F(4x4,3x3) kernel as they ran (10 passed) while it was part of the library - see
profiles/README.md, round 4.  Not collected by pytest: the kernel is not in the product library."""
# fused F(4x4,3x3): the transform matrices of F(4x4,3x3) (entries up to 8 and 1/24) at the short K of these layers
WINO4F_REL = 6e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout,G,film,res,ldx", [
    (1, 16, 32, 128, 64, 8, False, False, 0),    # ONE item: every side is padding (must stay 0 after the activation)
    (2, 32, 64, 128, 128, 8, True, True, 0),     # 2 x 2 patches x 2 slabs per image, FiLM, residual
    (1, 16, 32, 8, 64, 2, False, False, 0),      # two chunks: shorter than the pipeline's prefetch
    (1, 32, 32, 12, 64, 1, True, False, 0),      # three chunks
    (1, 16, 64, 20, 192, 1, False, True, 0),     # five chunks, three slabs
    (2, 16, 64, 256, 128, 8, True, True, 0),     # Cin = 256: 64 chunks
    (1, 32, 32, 128, 64, 8, True, True, 384),    # STRIDED input: a 128-channel slice at channel offset 128 of 384-float rows
    (16, 64, 128, 16, 128, 4, False, True, 0),   # 512 items on 256 persistent workgroups: a second item per workgroup
    (1, 16, 32, 512, 64, 8, True, False, 0),     # Cin = 512: the whole affine table
])
def test_gn_conv3x3_winograd4_fused_matches_torch(lib, device, B, H, W, Cin, Cout, G, film, res, ldx):
    """ResnetBlock `Block` = conv3x3(SiLU(FiLM(GroupNorm(x)))) through the fused Winograd F(4x4,3x3) kernel
    (kernels_wino4_fused.hip) against torch in fp64, with the GroupNorm statistics of the output it leaves."""
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
    if ldx:
        c0 = min(128, (ldx - Cin) // 4 * 4)
        wide = torch.full((B, H, W, ldx), 1e30, device=device)
        wide[..., c0:c0 + Cin] = xd
        xd = wide[..., c0:]
        assert xd.data_ptr() % 16 == 0
    xptr = C.c_void_p(xd.data_ptr())
    gd, bed, wd, bd = gamma.to(device), beta.to(device), w.to(device), b.to(device)
    ssd = ss.to(device) if film else None
    rd = r.permute(0, 2, 3, 1).contiguous().to(device) if res else None
    y = torch.full((B, H, W, Cout), float("nan"), device=device)
    Go = 8 if (Cout // 8) % 16 == 0 else (4 if (Cout // 4) % 16 == 0 else 0)
    ostats = torch.full((B, max(Go, 1), 2), float("nan"), device=device)
    call = lambda out: E.check(lib.kd_gn_conv3x3_winograd4_fused_nhwc(
        xptr, E.ptr(gd), E.ptr(bed), E.ptr(ssd) if film else None, E.ptr(wd), E.ptr(bd),
        E.ptr(rd) if res else None, E.ptr(out), B, H, W, Cin, Cout, G, 1e-5, None, ldx, E.current_stream()))
    call(y)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    err = float((got - ref).norm() / ref.norm())
    print(f"fused F(4x4,3x3) Cin {Cin} {H}x{W}: rel-L2 {err:.2e}, max {float((got - ref).abs().max() / ref.abs().max()):.2e}")
    assert err <= WINO4F_REL, err
    assert float((got - ref).abs().max()) <= 5e-5 * float(ref.abs().max()), "element-wise outlier"
    y2 = torch.empty_like(y)
    call(y2)
    assert torch.equal(y, y2)
    if Go and Cin % Go == 0 and (Cin // Go) % 4 == 0:   # output statistics over Go groups (the input's GroupNorm then uses Go groups as well)
        h2 = F.group_norm(x.double(), Go, gamma.double(), beta.double(), eps=1e-5)
        if film:
            h2 = h2 * (ss[:, :Cin, None, None].double() + 1) + ss[:, Cin:, None, None].double()
        ref2 = F.conv2d(F.silu(h2), w.double(), b.double(), padding=1) + (r.double() if res else 0)
        E.check(lib.kd_gn_conv3x3_winograd4_fused_nhwc(
            xptr, E.ptr(gd), E.ptr(bed), E.ptr(ssd) if film else None, E.ptr(wd), E.ptr(bd), E.ptr(rd) if res else None,
            E.ptr(y2), B, H, W, Cin, Cout, Go, 1e-5, E.ptr(ostats), ldx, E.current_stream()))
        grp = ref2.reshape(B, Go, -1)
        got_s = ostats.cpu().double()
        assert torch.allclose(got_s[..., 0], grp.mean(dim=-1), rtol=0, atol=1e-5 * float(ref2.abs().max()))
        assert torch.allclose(got_s[..., 1], (grp.var(dim=-1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-5, atol=0)


def test_gn_conv3x3_winograd4_fused_rejects_unsupported_shapes(lib, device):
    E = _E()
    t = torch.zeros(4096, device=device)
    p = E.ptr(t)
    for shape in [(1, 24, 32, 32, 64), (1, 16, 48, 32, 64), (1, 16, 32, 30, 64), (1, 16, 32, 32, 96), (1, 16, 32, 516, 64)]:
        rc = lib.kd_gn_conv3x3_winograd4_fused_nhwc(p, p, p, None, p, p, None, p, *shape, 2, 1e-5, None, 0, E.current_stream())
        assert rc != 0 and b"F(4x4,3x3)" in lib.kd_last_error(), (shape, lib.kd_last_error())


