"""Full-size parity: the HIP engine against the CPU oracle at the dims BASELINE.json's configs name
(no reduced `dim`), through the drop-in Python surface and the C ABI.

  C1  uncond base UNet, dim 256, 64x64, batch 1, T = 50, whole sampler   train_uncond.py:30-36, sample_uncond.py:49-55
  C2  seg-cond base UNet, dim 256, text + 4 label planes, one forward     train.py:30-39, sample_cond.py:36-48
  C3  SR UNet 64->256, dim 128: one forward and sampler steps, Winograd and direct plans   train_ultra_res.py:39-48
  C4  unet3 (blocks 2,4,6,8) on a 256x256 crop, one forward               train_ultra_res.py:51-60,88
      text + low-res SR UNet with cond_dim 512, one forward                train.py:42-53
      3-stage cascade chained 1 -> 2 -> 3 (reduced dims, full structure)   sample_ultra_res.py:264-270

Tolerances are the ones of tests/test_unet_gpu.py (2e-5 relative L2 for one forward, 2e-3 absolute on
sampled images).  The oracle runs on the host cores of the GPU box: a 343-459 GFLOP forward takes seconds.
"""
import time

import pytest
import torch

import helpers as H
from oracle import imagen_ref as R
from oracle import sampler_ref as RS

pytestmark = pytest.mark.gpu

FWD_REL_L2 = 2e-5
SAMPLE_ABS = 2e-3

F_, T_ = False, True
FULL_KW = {
    # train_uncond.py:30-36
    "uncond1": dict(dim=256, dim_mults=(1, 2, 4, 8), cond_dim=512, num_resnet_blocks=3,
                    layer_attns=(F_, T_, T_, T_), layer_cross_attns=(F_, T_, T_, T_)),
    # train.py:30-39
    "seg1": dict(dim=256, dim_mults=(1, 2, 3, 4), cond_dim=512, text_embed_dim=3, num_resnet_blocks=3,
                 layer_attns=(F_, T_, T_, T_), layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=4),
    # train.py:42-53
    "seg2": dict(dim=128, cond_dim=512, text_embed_dim=3, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2,
                 memory_efficient=True, layer_attns=(F_, F_, F_, T_), layer_cross_attns=(F_, F_, T_, T_),
                 init_conv_to_final_conv_residual=True, cond_images_channels=4),
    # train_ultra_res.py:39-48
    "ultra2": dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                   layer_attns=(F_, F_, F_, T_), layer_cross_attns=(F_, F_, T_, T_),
                   init_conv_to_final_conv_residual=True, cond_images_channels=3),
    # train_ultra_res.py:51-60
    "ultra3": dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True,
                   layer_attns=False, layer_cross_attns=(F_, F_, F_, T_), init_conv_to_final_conv_residual=True,
                   cond_images_channels=3),
}


def _oracle(name, lowres, seed, text=False, fast=True):
    kw = dict(FULL_KW[name])
    if not text:
        kw.pop("text_embed_dim", None)
        u = R.Unet(**kw, lowres_cond=lowres, cond_on_text=False, text_embed_dim=None)
    else:
        u = R.Unet(**kw, lowres_cond=lowres, cond_on_text=True)
    u = H.randomize_(u, seed).eval()
    # fast: the convolutions in channels_last (a third less host time; its other fp32 summation order moves the result by
    # ~1-3e-6 rel-L2, so the C3 fixture - whose errors DESIGN.md quotes - keeps the default layout)
    return H.fast_oracle(u) if fast else u


def _fwd_inputs(B, S, lowres, cc, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, S, S, generator=g)
    lr = torch.randn(B, 3, S, S, generator=g) if lowres else None
    cond = torch.rand(B, cc, S, S, generator=g) if cc else None
    t = torch.randn(B, generator=g) * 3
    tl = torch.full((B,), -1.3) if lowres else None
    return x, lr, cond, t, tl


def _dv(device):
    return lambda v: None if v is None else v.to(device)


# ------------------------------------------------------------------------------- C3: the headline UNet
@pytest.fixture(scope="module")
def c3():
    """Oracle results of the headline UNet, computed once for both plans (Winograd / direct)."""
    ou = _oracle("ultra2", True, seed=21, fast=False)
    B, S = 2, 256
    inp = _fwd_inputs(B, S, True, 3, seed=5)
    x, lr, cond, t, tl = inp
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    fwd_s = time.perf_counter() - t0
    # sampler: stage 2 of a cascade, T = 3 (t = 1 -> 2/3 -> 1/3 -> 0: first, middle and last-step branches)
    oim = RS.Imagen([R.NullUnet(), ou], image_sizes=(64, S), timesteps=(3, 3), pred_objectives=("noise", "noise"),
                    condition_on_text=False)
    g = torch.Generator().manual_seed(6)
    start = torch.rand(1, 3, 64, 64, generator=g)
    scond = torch.rand(1, 3, S, S, generator=g)
    nf = RS.generator_noise_fn(31)
    trace = []
    sref = oim.sample(noise_fn=nf, batch_size=1, cond_images=scond, start_image_or_video=start,
                      start_at_unet_number=2, trace=trace)
    return dict(ou=ou, inp=inp, ref=ref, fwd_s=fwd_s, oim=oim, start=start, scond=scond, nf=nf, sref=sref,
                trace=trace)


@pytest.mark.parametrize("conv_algo", [0, 1])
def test_c3_sr_unet_forward_and_sampler_steps_match_oracle(device, c3, conv_algo):
    import imagen_pytorch as ip

    dv = _dv(device)
    pu = H.product_unet_like(c3["ou"]).to(device)
    pu.conv_algo = conv_algo
    x, lr, cond, t, tl = c3["inp"]
    got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
    err = H.rel_l2(got, c3["ref"])
    assert err < FWD_REL_L2, f"C3 forward, conv_algo={conv_algo}: rel-L2 {err:.3e}"
    # full p_sample steps (x0, dynamic threshold, posterior, noise) on the same UNet
    pim = ip.Imagen([ip.NullUnet(), pu], image_sizes=(64, 256), timesteps=(3, 3), pred_objectives=("noise", "noise"),
                    condition_on_text=False).to(device)
    pim.unets[1].conv_algo = conv_algo
    ptrace = []
    sgot = pim.sample(noise_fn=c3["nf"], batch_size=1, cond_images=dv(c3["scond"]),
                      start_image_or_video=dv(c3["start"]), start_at_unet_number=2, trace=ptrace, device=device)
    for k, (a, b) in enumerate(zip(ptrace, c3["trace"])):
        assert H.rel_l2(a, b) < 1e-4 * (k + 1), (k, H.rel_l2(a, b))
    assert (sgot.cpu() - c3["sref"]).abs().max() < SAMPLE_ABS


def test_c3_sr_unet_at_the_benchmark_batch_matches_oracle(device, c3):
    """The plan bench.py times: batch 16.  With the position GEMMs on the bf16 pipe all 56 ResnetBlock 3x3 convs (Cin >= 128)
    run as Winograd F(4x4,3x3); with fp32 MFMA GEMMs the 31 with Cin >= 512 (the 32x32 and 16x16 levels) do and the other 25
    stay on the fused F(2x2,3x3) kernel (at batch 2, above, the deep levels do not fill whole 128-row tile slabs and stay
    on F(2x2,3x3)).  One forward against the oracle (7.3 TFLOP on the host)."""
    import ctypes as C
    from imagen_pytorch import _engine as E

    ou = c3["ou"]
    B, S = 16, 256
    x, lr, cond, t, tl = _fwd_inputs(B, S, True, 3, seed=15)
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    dv = _dv(device)
    errs, outs = {}, {}
    # default plan (the F(4x4,3x3) position GEMMs on the bf16 matrix pipe as three-piece fp32 products; V of the 64 x 64 level
    # and above in fp32, of the others as planes), the same with V as planes everywhere, with fp32 token GEMMs, with fp32 MFMA
    # position GEMMs (default threshold, and the same 56 layers), and the plan without F(4x4,3x3)
    nlin = {}
    for w43, x3, lin in ((0, 0, 0), (0, 1, 0), (0, 0, -1), (0, -1, 0), (128, -1, 0), (-1, 0, 0)):
        pu = H.product_unet_like(ou).to(device)
        pu.wino43_min_cin = w43
        pu.gemm_bf16x3 = x3
        pu.x3_linear = lin
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
        key = (w43, x3) if lin == 0 else "fp32 token GEMMs"
        errs[key] = H.rel_l2(got, ref)
        outs[key] = got
        buf = C.create_string_buffer(1 << 20)
        E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
        labels = buf.value.decode()
        n4 = {(0, 0): 56, (0, 1): 56, (0, -1): 31, (128, -1): 56, (-1, 0): 0}[w43, x3]
        assert labels.count("wino4 gemm") == n4, labels.count("wino4 gemm")
        assert labels.count("wino4 gemm bf16x3") == (n4 if x3 >= 0 else 0), labels.count("wino4 gemm bf16x3")
        assert labels.count("wino fused") == 56 - n4, labels.count("wino fused")
        # V as fp32 (split by the GEMM's loader waves) on the 28 layers of the 64 x 64 level and above, as planes on the other 28
        assert labels.count("wino4_in3 M") == {0: 28, 1: 56}.get(x3, 0) * (w43 == 0), labels.count("wino4_in3 M")
        # the attention projections and the feed-forward of the 16 x 16 / 32 x 32 levels (K >= 512) on the bf16x3 kernel's
        # epilogue form - unless switched off, or the bf16x3 kernels are off altogether
        nlin[key] = labels.count("conv k1 x3 M") + labels.count("conv k2 x3 M")
        assert (nlin[key] >= 20) == (lin == 0 and x3 >= 0), (key, nlin[key])
        del pu
    # the two forms of V are the same numbers: the loader waves split what the transform would have split
    assert torch.equal(outs[0, 0], outs[0, 1])
    # the default plan with its F(4x4,3x3) layers and its 1x1 convs on the bf16x3 kernel in sets of 8 images (what the plan does
    # by itself where the maps of the whole batch pass 4 GB: unet3's outer levels at batch 8): the whole-batch result to fp32
    # rounding, twice the launches for those layers
    pu = H.product_unet_like(ou).to(device)
    pu.wino4_max_images = 8
    got8 = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
    buf = C.create_string_buffer(1 << 20)
    E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
    labels8 = buf.value.decode()
    # (the 16 layers of the 16 x 16 level: 128 tiles per set, half a 256-row tile of the bf16x3 kernel - their GEMMs on fp32 MFMA)
    assert labels8.count("wino4 gemm") == 2 * 56 and labels8.count("wino4 gemm bf16x3") == 2 * 40, (
        labels8.count("wino4 gemm"), labels8.count("wino4 gemm bf16x3"))
    n_conv8 = labels8.count("conv k1 x3 M")
    assert n_conv8 > nlin[0, 0] - 8, (n_conv8, nlin[0, 0])   # (the convs doubled; the token GEMMs - flat rows - as they were)
    e8 = H.rel_l2(got8, outs[0, 0])
    print(f"C3 plan at batch 16 in sets of 8 images: {n_conv8} 1x1-conv / token-GEMM launches on bf16x3, rel-L2 against the whole-batch plan {e8:.2e}")
    assert e8 < 2e-6 and H.rel_l2(got8, ref) < FWD_REL_L2
    del pu
    print(f"C3 plan at batch 16: {nlin[0, 0]} token GEMMs / 1x1 convs on bf16x3; rel-L2 {errs[0, 0]:.3e} with them, "
          f"{errs['fp32 token GEMMs']:.3e} with conv_buf_kernel (fp32 MFMA)")
    assert errs[0, 0] < 1.5 * errs["fp32 token GEMMs"] + 1e-7, errs   # fp32-class products: nothing is lost
    print(f"C3 forward at batch 16: rel-L2 {errs[0, 0]:.3e} with F(4x4,3x3) on all 56 layers (bf16x3 GEMMs), "
          f"{errs[128, -1]:.3e} the same with fp32 MFMA GEMMs, {errs[0, -1]:.3e} on 31 layers, {errs[-1, 0]:.3e} without F(4x4,3x3)")
    assert all(e < FWD_REL_L2 for e in errs.values()), errs
    assert errs[0, 0] < 1.5 * errs[128, -1] + 1e-7, errs   # (the bf16x3 products themselves cost nothing)


# ------------------------------------------------------------------------------- C1: end to end
def test_c1_uncond_base_unet_50_steps_end_to_end(device):
    """BASELINE configs[0]: unconditional base UNet 64x64, 50 DDPM steps, batch 1 - the whole sampler on both
    paths with identical injected noise (BASELINE.md §3 'full-parity config')."""
    import imagen_pytorch as ip

    ou = _oracle("uncond1", False, seed=41)
    oim = RS.Imagen([ou], image_sizes=(64,), timesteps=(50,), pred_objectives=("noise",), condition_on_text=False)
    pim = ip.Imagen([ip.Unet(**oim.unets[0]._locals)], image_sizes=(64,), timesteps=(50,),
                    pred_objectives=("noise",), condition_on_text=False)
    pim.load_state_dict(oim.state_dict(), strict=True)
    pim = pim.to(device)
    nf = RS.generator_noise_fn(2024)
    otrace, ptrace = [], []
    ref = oim.sample(noise_fn=nf, batch_size=1, trace=otrace)
    got = pim.sample(noise_fn=nf, batch_size=1, trace=ptrace, device=device)
    assert len(otrace) == len(ptrace) == 50
    worst = max(H.rel_l2(a, b) for a, b in zip(ptrace, otrace))
    err = float((got.cpu() - ref).abs().max())
    print(f"C1 end to end: max|diff| {err:.3e}, worst per-step rel-L2 {worst:.3e}")
    assert err < SAMPLE_ABS, err
    # the production path (graph replay of the whole loop) gives the same image as the stepwise trace run
    got2 = pim.sample(noise_fn=nf, batch_size=1, device=device)
    assert torch.equal(got, got2)


# ------------------------------------------------------------------------------- C2: seg-cond base UNet
def test_c2_segcond_base_unet_forward_matches_oracle(device):
    """BASELINE configs[1] UNet at full size: text_embeds (B,1,3), four one-hot label planes
    (sample_cond.py:36-38, 75-80)."""
    ou = _oracle("seg1", False, seed=51, text=True)
    pu = H.product_unet_like(ou).to(device)
    B, S = 2, 64
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g) * 2
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    mask = torch.any(text != 0.0, dim=-1)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, 1024, 1024), generator=g), 4)
    labels = labels.permute(0, 3, 1, 2).float()   # resized inside (nearest), as the reference passes them
    with torch.no_grad():
        ref = ou(x, t, text_embeds=text, text_mask=mask, cond_images=labels)
    dv = _dv(device)
    got = pu(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask), cond_images=dv(labels))
    err = H.rel_l2(got, ref)
    assert err < FWD_REL_L2, f"C2 forward: rel-L2 {err:.3e}"


def _plan_labels(pu, B, S, device, with_text):
    import ctypes as C
    from imagen_pytorch import _engine as E

    buf = C.create_string_buffer(1 << 20)
    E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=with_text), 1, buf, len(buf), E.current_stream()))
    return buf.value.decode()


def test_c2_segcond_base_unet_at_batch_16_forward_and_guided_sampler_match_oracle(device):
    """BASELINE configs[1] on the plan it actually runs (train.py:30-39, sample_cond.py:40-48: batch 16): at batch 16
    the 16x16 level (dim 768) and the 8x8 middle (dim 1024) go to Winograd F(4x4,3x3) with K = 768 / 1536 / 1024 - shapes
    the batch-2 test above never builds.  One forward (4.2 TFLOP on the host) and two text-conditioned sampler steps
    per guidance scale (cond_scale 1 and 2.5: sample.py:51-60) at the reference's dims against the oracle."""
    import imagen_pytorch as ip

    ou = _oracle("seg1", False, seed=53, text=True)
    B, S = 16, 64
    g = torch.Generator().manual_seed(28)
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g) * 2
    text, mask = _text_inputs(B)
    text[3, 0] = torch.tensor([0.3, -0.2, 1.0])   # one sample with another text row
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, S, S), generator=g), 4).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        ref = ou(x, t, text_embeds=text, text_mask=mask, cond_images=labels)
    dv = _dv(device)
    pu = H.product_unet_like(ou).to(device)
    got = pu(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask), cond_images=dv(labels))
    err = H.rel_l2(got, ref)
    plan = _plan_labels(pu, B, S, device, True)
    n4 = plan.count("wino4 gemm")
    print(f"C2 forward at batch 16: rel-L2 {err:.3e}; {n4} F(4x4,3x3) layers, {plan.count('wino fused')} fused F(2x2,3x3)")
    assert n4 >= 8, n4     # the 16x16 level's ResnetBlocks at the least
    assert err < FWD_REL_L2, f"C2 forward at batch 16: rel-L2 {err:.3e}"
    del pu
    # guided sampler at full dims: T = 2 (first and last-step branches), the loop replayed from the captured graph
    kw = dict(image_sizes=(S,), timesteps=(2,), pred_objectives=("noise",), text_embed_dim=3)
    oim = RS.Imagen([ou], **kw)
    pim = ip.Imagen([ip.Unet(**ou._locals)], **kw)
    pim.load_state_dict(oim.state_dict(), strict=True)
    pim = pim.to(device)
    nf = RS.generator_noise_fn(29)
    for cs in (1.0, 2.5):
        sref = oim.sample(noise_fn=nf, text_embeds=text, cond_images=labels, cond_scale=cs)
        sgot = pim.sample(noise_fn=nf, text_embeds=dv(text), cond_images=dv(labels), cond_scale=cs, device=device)
        d = float((sgot.cpu() - sref).abs().max())
        print(f"C2 guided sampler, cond_scale {cs}: max|diff| {d:.3e}")
        assert d < SAMPLE_ABS, (cs, d)


def test_ultra_unet1_at_batch_16_matches_oracle(device):
    """train_ultra_res.py:29-36 (configs[3] stage 1) at batch 16: its 32x32 (dim 512) and 16x16 (dim 1024) levels and the
    8x8 middle (dim 2048) switch to F(4x4,3x3) from batch 8 on; one forward against the oracle (5.5 TFLOP on the host)."""
    u1 = dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
              layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=3)
    ou = H.fast_oracle(H.randomize_(R.Unet(**u1, cond_on_text=False, text_embed_dim=None), 84).eval())
    B, S = 16, 64
    x, _, cond, t, _ = _fwd_inputs(B, S, False, 3, seed=19)
    with torch.no_grad():
        ref = ou(x, t, cond_images=cond)
    dv = _dv(device)
    pu = H.product_unet_like(ou).to(device)
    got = pu(dv(x), dv(t), cond_images=dv(cond))
    err = H.rel_l2(got, ref)
    plan = _plan_labels(pu, B, S, device, False)
    n4 = plan.count("wino4 gemm")
    print(f"ultra unet1 forward at batch 16: rel-L2 {err:.3e}; {n4} F(4x4,3x3) layers, {plan.count('wino fused')} fused")
    assert n4 >= 16, n4
    assert err < FWD_REL_L2, err


def test_text_lowres_sr_unet_cond_dim_512_forward_matches_oracle(device):
    """train.py:42-53: the SR UNet of the seg-cond cascade - text + low-res conditioning + cond_dim 512 (the
    cross-attention context is 4 time tokens + 36 pooled text tokens of width 512)."""
    ou = _oracle("seg2", True, seed=52, text=True)
    pu = H.product_unet_like(ou).to(device)
    B, S = 1, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 3, S, S, generator=g)
    lr = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g)
    tl = torch.full((B,), -1.1)
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    mask = torch.any(text != 0.0, dim=-1)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, S, S), generator=g), 4).permute(0, 3, 1, 2).float()
    dv = _dv(device)
    for drop in (0.0, 1.0):
        with torch.no_grad():
            ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, text_embeds=text, text_mask=mask,
                     cond_images=labels, cond_drop_prob=drop)
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), text_embeds=dv(text),
                 text_mask=dv(mask), cond_images=dv(labels), cond_drop_prob=drop)
        err = H.rel_l2(got, ref)
        assert err < FWD_REL_L2, f"text+lowres SR UNet, drop={drop}: rel-L2 {err:.3e}"


# ------------------------------------------------------------------------------- the remaining reference kwargs sets
def _text_inputs(B):
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)   # sample_cond.py:37
    return text, torch.any(text != 0.0, dim=-1)


def test_kumar_two_stage_unets_forward_match_oracle(device):
    """train_kumar.py:27-81: base UNet with text + ONE conditioning channel, and an SR UNet that is built WITHOUT
    text_embed_dim and gets its text conditioning from `Imagen(text_embed_dim=3)` re-casting it
    (condition_on_text defaults to True); `timesteps=1000` as a scalar for both stages."""
    import imagen_pytorch as ip

    k1 = dict(dim=256, dim_mults=(1, 2, 3, 4), cond_dim=512, text_embed_dim=3, num_resnet_blocks=3,
              layer_attns=(F_, T_, T_, T_), layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=1)   # :29-38
    k2 = dict(dim=128, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
              layer_attns=(F_, F_, F_, T_), layer_cross_attns=(F_, F_, T_, T_), init_conv_to_final_conv_residual=True,
              cond_images_channels=1)                                                                       # :41-51
    oim = RS.Imagen([R.Unet(**k1), R.Unet(**k2)], image_sizes=(64, 256), timesteps=1000, text_embed_dim=3)
    for n, u in enumerate(oim.unets):
        H.fast_oracle(H.randomize_(u, 91 + n).eval())
    pim = ip.Imagen([ip.Unet(**k1), ip.Unet(**k2)], image_sizes=(64, 256), timesteps=1000, text_embed_dim=3,
                    random_crop_sizes=(None, None))
    pim.load_state_dict(oim.state_dict(), strict=True)   # same key set after the re-cast (text modules of the SR UNet)
    pim = pim.to(device)
    assert [s.num_timesteps for s in pim.noise_schedulers] == [1000, 1000]
    assert pim.unets[1].cond_on_text and pim.unets[1].lowres_cond and not pim.unets[0].lowres_cond
    dv = _dv(device)
    g = torch.Generator().manual_seed(17)
    for stage, (B, S) in enumerate(((2, 64), (1, 256))):
        ou, pu = oim.unets[stage], pim.unets[stage]
        x = torch.randn(B, 3, S, S, generator=g)
        t = torch.randn(B, generator=g) * 2
        cond = torch.rand(B, 1, 1024 if stage == 0 else S, 1024 if stage == 0 else S, generator=g)
        lr = torch.randn(B, 3, S, S, generator=g) if stage else None
        tl = torch.full((B,), -1.2) if stage else None
        text, mask = _text_inputs(B)
        with torch.no_grad():
            ref = ou(x, t, text_embeds=text, text_mask=mask, cond_images=cond, lowres_cond_img=lr, lowres_noise_times=tl)
        got = pu(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask), cond_images=dv(cond), lowres_cond_img=dv(lr),
                 lowres_noise_times=dv(tl))
        err = H.rel_l2(got, ref)
        assert err < FWD_REL_L2, f"kumar unet{stage + 1}: rel-L2 {err:.3e}"


def test_segcond_unet3_blocks_2444_text_forward_matches_oracle(device):
    """train.py:55-65: the third UNet of the seg-cond cascade - text, low-res, cond_dim 512, resnet blocks
    (2, 4, 4, 4), FOUR conditioning channels - on the 256 crop it is trained on (train.py:92)."""
    kw = dict(dim=128, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 4, 4), memory_efficient=True,
              layer_attns=False, layer_cross_attns=(F_, F_, F_, T_), init_conv_to_final_conv_residual=True,
              cond_images_channels=4)
    ou = H.fast_oracle(H.randomize_(R.Unet(**kw, lowres_cond=True, cond_on_text=True, text_embed_dim=3), 95).eval())
    pu = H.product_unet_like(ou).to(device)
    B, S = 1, 256
    g = torch.Generator().manual_seed(18)
    x = torch.randn(B, 3, S, S, generator=g)
    lr = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g)
    tl = torch.full((B,), -0.9)
    text, mask = _text_inputs(B)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, S, S), generator=g), 4).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, text_embeds=text, text_mask=mask, cond_images=labels)
    dv = _dv(device)
    got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), text_embeds=dv(text), text_mask=dv(mask),
             cond_images=dv(labels))
    err = H.rel_l2(got, ref)
    assert err < FWD_REL_L2, f"train.py unet3: rel-L2 {err:.3e}"


# ------------------------------------------------------------------------------- C4 stage 3 on a crop
def test_unet3_on_a_256_crop_forward_matches_oracle(device):
    """train_ultra_res.py:51-60 at full dim on the 256x256 crop it is trained on (:88): 753.5 GFLOP."""
    from imagen_pytorch import _engine as E

    ou = _oracle("ultra3", True, seed=61)
    pu = H.product_unet_like(ou).to(device)
    x, lr, cond, t, tl = _fwd_inputs(1, 256, True, 3, seed=10)
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    dv = _dv(device)
    got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
    err = H.rel_l2(got, ref)
    assert err < FWD_REL_L2, f"unet3 crop forward: rel-L2 {err:.3e}"
    gmac = E.load().kd_unet_macs(pu.engine(1, 256, device, with_text=False)) / 1e9
    assert abs(gmac - 376.7) / 376.7 < 0.01, gmac


def test_c4_full_size_cascade_64_256_1024_matches_oracle(device):
    """BASELINE configs[3]: the full 3-stage cascade 64 -> 256 -> 1024 at the reference's dims
    (train_ultra_res.py:29-60), one sample() call through all three UNets with the hipGraph-replayed inner loop,
    against the oracle: batch 1, two timesteps per stage (26 TFLOP on the host cores).  The batch-8 run of the
    same cascade is checked through properties below (the oracle cannot run it in a test)."""
    import imagen_pytorch as ip

    u1 = dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
              layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=3)   # train_ultra_res.py:29-36
    ous = [H.fast_oracle(H.randomize_(R.Unet(**u1, cond_on_text=False, text_embed_dim=None), 81).eval()), _oracle("ultra2", True, 82),
           _oracle("ultra3", True, 83)]
    kw = dict(image_sizes=(64, 256, 1024), timesteps=(2, 2, 2), pred_objectives=("noise", "noise", "noise"),
              condition_on_text=False)
    oim = RS.Imagen(ous, **kw)
    pim = ip.Imagen([ip.Unet(**u._locals) for u in oim.unets], random_crop_sizes=(None, None, 256), **kw)
    pim.load_state_dict(oim.state_dict(), strict=True)
    del ous
    pim = pim.to(device)
    g = torch.Generator().manual_seed(14)
    cond = torch.rand(1, 3, 1024, 1024, generator=g)
    nf = RS.generator_noise_fn(7)
    ref = oim.sample(noise_fn=nf, batch_size=1, cond_images=cond, return_all_unet_outputs=True)
    got = pim.sample(noise_fn=nf, batch_size=1, cond_images=cond.to(device), return_all_unet_outputs=True, device=device)
    for stage, (a, b) in enumerate(zip(got, ref), 1):
        err = float((a.cpu() - b).abs().max())
        print(f"C4 cascade stage {stage}: max|diff| {err:.3e}")
        assert err < SAMPLE_ABS * stage, (stage, err)
    del oim, ref
    # batch 8 (configs[3]'s batch) on the engine: seeded, deterministic, in range; every sample differs
    a = pim.sample(batch_size=8, cond_images=cond.to(device).expand(8, -1, -1, -1), seed=3, device=device)
    b = pim.sample(batch_size=8, cond_images=cond.to(device).expand(8, -1, -1, -1), seed=3, device=device)
    assert a.shape == (8, 3, 1024, 1024) and torch.equal(a, b) and a.min() >= 0 and a.max() <= 1
    assert all(not torch.equal(a[0], a[i]) for i in range(1, 8))


# ------------------------------------------------------------------------------- 3-stage cascade, chained
def test_three_stage_cascade_chained_matches_oracle(device):
    """BASELINE configs[3] structure (64 -> 256 -> 1024 cascade, one sample() call through all three UNets,
    each stage conditioned on the previous one's output) at reduced dims / sizes so the oracle runs in
    seconds: unet kwargs of train_ultra_res.py:29-60 at dim 32, image sizes (16, 32, 64)."""
    import imagen_pytorch as ip

    ous = [H.oracle_unet("ultra1", seed=71), H.oracle_unet("ultra2", lowres_cond=True, seed=72),
           H.oracle_unet("ultra3", lowres_cond=True, seed=73)]
    kw = dict(image_sizes=(16, 32, 64), timesteps=(4, 3, 3), pred_objectives=("noise", "v", "v"),
              condition_on_text=False)
    oim = RS.Imagen(ous, **kw)
    pim = ip.Imagen([ip.Unet(**u._locals) for u in oim.unets], random_crop_sizes=(None, None, 16), **kw)
    pim.load_state_dict(oim.state_dict(), strict=True)
    pim = pim.to(device)
    g = torch.Generator().manual_seed(12)
    cond = torch.rand(2, 3, 64, 64, generator=g)
    nf = RS.generator_noise_fn(99)
    ref = oim.sample(noise_fn=nf, batch_size=2, cond_images=cond, return_all_unet_outputs=True)
    got = pim.sample(noise_fn=nf, batch_size=2, cond_images=cond.to(device), return_all_unet_outputs=True,
                     device=device)
    assert [tuple(o.shape) for o in got] == [(2, 3, 16, 16), (2, 3, 32, 32), (2, 3, 64, 64)]
    for stage, (a, b) in enumerate(zip(got, ref), 1):
        err = float((a.cpu() - b).abs().max())
        assert err < SAMPLE_ABS * stage, (stage, err)   # each stage inherits the previous stage's deviation
    # stop_at_unet_number / start_at_unet_number split of the same chain (the reference's per-stage loop,
    # sample_ultra_res.py:264-270) gives the same images as the single call
    s1 = pim.sample(noise_fn=nf, batch_size=2, cond_images=cond.to(device), stop_at_unet_number=1, device=device)
    assert torch.equal(s1, got[0])
    s2 = pim.sample(noise_fn=nf, batch_size=2, cond_images=cond.to(device), start_at_unet_number=2,
                    stop_at_unet_number=2, start_image_or_video=s1, device=device)
    assert torch.equal(s2, got[1])
