"""GPU parity of the whole UNet forward and of the sampler loop against the CPU oracle, through
the drop-in Python surface (which calls the C ABI).  Same weights, same inputs, same noise."""
import pytest
import torch

import helpers as H
from oracle import sampler_ref as RS

pytestmark = pytest.mark.gpu

# Stated fp32 tolerances (north_star: "within a stated fp32 tolerance").
FWD_REL_L2 = 2e-5        # one UNet forward: relative L2 of the output vs the CPU fp32 oracle
SAMPLE_ABS = 2e-3        # images in [0,1] after T denoising steps (error compounds through the loop)


def _inputs(name, B, S, lowres, seed=3):
    g = torch.Generator().manual_seed(seed)
    kw = H.UNET_KW[name]
    x = torch.randn(B, 3, S, S, generator=g)
    lr = torch.randn(B, 3, S, S, generator=g) if lowres else None
    cc = kw.get("cond_images_channels", 0)
    cond = torch.rand(B, cc, 2 * S, 2 * S, generator=g) if cc else None   # resized inside (nearest)
    t = torch.randn(B, generator=g) * 3
    tl = torch.full((B,), -1.3) if lowres else None
    return x, lr, cond, t, tl


@pytest.mark.parametrize("name,lowres,B,S", [
    ("small1", False, 2, 16),
    ("small2", True, 2, 32),
    ("ultra1", False, 2, 32),     # train_ultra_res.py:29-36 at dim 32 (cond images, attention at 3 levels)
    ("ultra1", False, 8, 64),     # same, 1024 / 256 / 64 tokens x 8 heads x 8: the matrix-core attention kernel
    ("ultra2", True, 3, 64),      # train_ultra_res.py:39-48 (memory efficient SR unet)  <- headline UNet
    ("ultra3", True, 1, 64),      # train_ultra_res.py:51-60 (blocks 2,4,6,8, no self-attention but mid)
    ("uncond1", False, 1, 32),    # train_uncond.py:30-36 (cond_dim 64 here)
])
def test_unet_forward_matches_oracle(device, name, lowres, B, S):
    ou = H.oracle_unet(name, lowres_cond=lowres, seed=11).eval()
    pu = H.product_unet_like(ou).to(device)
    x, lr, cond, t, tl = _inputs(name, B, S, lowres)
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    dv = lambda v: None if v is None else v.to(device)
    got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = H.rel_l2(got, ref)
    assert err < FWD_REL_L2, f"{name}: rel-L2 {err:.3e}"
    # second call reuses the cached plan and must be bit-identical (no stale state in the workspace)
    got2 = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
    assert torch.equal(got, got2)


def test_unet_forward_with_winograd_levels_matches_oracle(device):
    """The plan's Winograd F(2x2,3x3) path (deep ResnetBlock convs; GroupNorm+FiLM+SiLU fused into the
    input transform).  conv_algo=32 lowers the Cin threshold so the small test UNet exercises it on
    its 16x16 and 8x8 levels; conv_algo=1 (direct only) must agree to the same tolerance."""
    import ctypes as C
    from imagen_pytorch import _engine as E

    B, S = 4, 128   # (levels 32 @ 64^2, 64 @ 32^2, 128 @ 16^2, 256 @ 8^2: the fused kernel takes Cout % 128 on maps of 8 x 16 multiples)
    ou = H.oracle_unet("ultra2", lowres_cond=True, seed=13).eval()
    x, lr, cond, t, tl = _inputs("ultra2", B, S, True, seed=8)
    with torch.no_grad():
        ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    dv = lambda v: None if v is None else v.to(device)
    import os

    outs, n_gemm = {}, {}
    # (conv_algo, wino_slice_mb): the third variant walks every Winograd layer in 256-tile slices
    # conv_algo=3: the FUSED Winograd kernel (kernels_wino_fused128.hip) wherever its shape rules allow, even
    # where the launch would not fill the chip (here the 16x16 maps with 128 output channels)
    for algo, slice_mb in ((32, None), (1, None), (32, "1"), (3, None)):
        pu = H.product_unet_like(ou).to(device)
        pu.conv_algo = algo
        if slice_mb is not None:
            pu.wino_slice_mb = int(slice_mb)   # plan option (kd_unet_config_t::wino_slice_mb), not an environment switch
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
        err = H.rel_l2(got, ref)
        assert err < FWD_REL_L2, f"conv_algo={algo} slice={slice_mb}: rel-L2 {err:.3e}"
        buf = C.create_string_buffer(1 << 20)
        E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=False), 1, buf, len(buf),
                                         E.current_stream()))
        n_wino = buf.value.decode().count("wino gemm")
        n_fused = buf.value.decode().count("wino fused")
        if algo != 3:   # 3 keeps the batched-GEMM path from Cin >= 256 like 0
            assert (n_wino > 0) == (algo == 32), f"conv_algo={algo}: {n_wino} Winograd GEMMs in the plan"
        assert (n_fused > 0) == (algo == 3), f"conv_algo={algo}: {n_fused} fused Winograd convs in the plan"
        outs[(algo, slice_mb)], n_gemm[(algo, slice_mb)] = got, n_wino
    assert H.rel_l2(outs[(32, None)], outs[(1, None)]) < FWD_REL_L2
    assert H.rel_l2(outs[(3, None)], outs[(1, None)]) < FWD_REL_L2
    assert n_gemm[(32, "1")] > n_gemm[(32, None)], "slicing did not split the Winograd layers"
    assert torch.equal(outs[(32, "1")], outs[(32, None)]), "sliced and unsliced Winograd must be bit-identical"


def test_unet_forward_with_winograd_f4_levels_matches_oracle(device):
    """conv_algo = 4: every ResnetBlock 3x3 conv whose shape fits runs as Winograd F(4x4,3x3) (36 batched GEMMs, GroupNorm /
    FiLM / SiLU in the input transform, bias / residual / next layer's GroupNorm partials in the output transform); the
    default plan takes that form for Cin >= 512 only, which no reduced-dim model reaches.  ultra2 / ultra3 at dim 32,
    batch 16, 128 x 128: the 32 x 32 (C = 64) and 16 x 16 (C = 128) levels qualify - skip concats, cross-attention
    blocks and the blocks whose residual rides in the output transform included."""
    import ctypes as C
    from imagen_pytorch import _engine as E

    for name in ("ultra2", "ultra3"):
        B, S = 16, 128
        ou = H.oracle_unet(name, lowres_cond=True, seed=23).eval()
        x, lr, cond, t, tl = _inputs(name, B, S, True, seed=9)
        with torch.no_grad():
            ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
        dv = lambda v: None if v is None else v.to(device)
        outs = {}
        for algo in (4, 1):
            pu = H.product_unet_like(ou).to(device)
            pu.conv_algo = algo
            got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
            err = H.rel_l2(got, ref)
            assert err < FWD_REL_L2, f"{name} conv_algo={algo}: rel-L2 {err:.3e}"
            outs[algo] = got.clone()   # (kd_unet_profile runs the plan again on the pointers of the last forward: `got` is its output)
            buf = C.create_string_buffer(1 << 20)
            E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
            n4 = buf.value.decode().count("wino4 gemm")
            assert (n4 >= 8) == (algo == 4), f"{name} conv_algo={algo}: {n4} F(4x4,3x3) layers in the plan"
            print(f"{name} conv_algo={algo}: rel-L2 vs oracle {err:.2e} ({n4} F(4x4,3x3) layers)")
            if algo == 4:
                n4_whole = n4
        assert H.rel_l2(outs[4], outs[1]) < FWD_REL_L2
        # the same layers in sets of 8 images (what the default plan does where V / D of the whole batch pass 4 GB - unet3's
        # outer levels at batch 8): two sets of launches per layer over one set's V and D, every per-image pointer (map,
        # statistics, FiLM rows, residual, output partials) moved on by a set - the whole-batch result to fp32 rounding
        pu = H.product_unet_like(ou).to(device)
        pu.conv_algo = 4
        pu.wino4_max_images = 8
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
        buf = C.create_string_buffer(1 << 20)
        E.check(E.load().kd_unet_profile(pu.engine(B, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
        n4 = buf.value.decode().count("wino4 gemm")
        assert n4 == 2 * n4_whole, (n4, n4_whole)
        assert H.rel_l2(got, outs[4]) < 2e-6 and H.rel_l2(got, ref) < FWD_REL_L2, (H.rel_l2(got, outs[4]), H.rel_l2(got, ref))


def test_conditioning_table_gives_bit_identical_samples(device):
    """The time conditioning of a step (embeddings, FiLM scale / shift, time tokens, their cross-attention K / V) depends
    on the schedule index alone when there is no text: the sampler computes it once per schedule into a table and an
    iteration restores its row with one gather (include/kd_engine.h: kd_sample_args_t::cond_table).  Same kernels, same
    inputs: samples must be bit-identical with the table off - base and low-res-conditioned UNets, with inpainting
    resampling (two iterations per schedule step), graph and eager - and a second call (table reused), a call with another
    low-res level (table rebuilt) and a call with another number of steps must each equal their table-off run."""
    import imagen_pytorch as ip

    _, pim = _imagen_pair(device, ["ultra1", "ultra2"], (16, 32), (5, 4), ("noise", "v"))
    g = torch.Generator().manual_seed(21)
    cond = torch.rand(2, 3, 32, 32, generator=g).to(device)
    low = torch.rand(2, 3, 16, 16, generator=g).to(device)
    ipt = torch.rand(2, 3, 32, 32, generator=g).to(device)
    msk = torch.zeros(2, 32, 32, device=device)
    msk[:, :8] = 1

    def run(table, **kw):
        pim.cond_table = 0 if table else -1
        return pim.sample(batch_size=2, cond_images=cond, device=device, seed=11, **kw)

    cases = [dict(stop_at_unet_number=1),
             dict(stop_at_unet_number=1, use_graph=False),
             dict(start_at_unet_number=2, start_image_or_video=low),
             dict(start_at_unet_number=2, start_image_or_video=low, inpaint_images=ipt, inpaint_masks=msk, inpaint_resample_times=2),
             dict(start_at_unet_number=2, start_image_or_video=low),                                   # table reused
             dict(start_at_unet_number=2, start_image_or_video=low, lowres_sample_noise_level=0.35),   # other level: rebuilt
             dict(start_at_unet_number=2, start_image_or_video=low)]
    for n, kw in enumerate(cases):
        a, b = run(True, **kw), run(False, **kw)
        assert torch.isfinite(a).all() and torch.equal(a, b), (n, float((a - b).abs().max()))
    assert not torch.equal(run(True, **cases[5]), run(True, **cases[6])), "the low-res level must matter"
    # a traced run walks the schedule one step per call: the table's rows are built on demand, two schedule steps (= the
    # batch) per run of the conditioning ops, the last chunk of the odd-length schedule half filled
    ta, tb = [], []
    a = run(True, trace=ta, start_at_unet_number=2, start_image_or_video=low, lowres_sample_noise_level=0.3)
    b = run(False, trace=tb, start_at_unet_number=2, start_image_or_video=low, lowres_sample_noise_level=0.3)
    assert torch.equal(a, b) and len(ta) == len(tb) == 4 and all(torch.equal(x, y) for x, y in zip(ta, tb))
    a3 = pim.sample(batch_size=3, cond_images=cond[:1].expand(3, -1, -1, -1).contiguous(), device=device, seed=5, stop_at_unet_number=1)
    pim.cond_table = -1
    b3 = pim.sample(batch_size=3, cond_images=cond[:1].expand(3, -1, -1, -1).contiguous(), device=device, seed=5, stop_at_unet_number=1)
    pim.cond_table = 0
    assert torch.equal(a3, b3)   # batch 3, T = 5: chunks of three schedule steps
    # another schedule length on the same plan
    pim2 = ip.Imagen([pim.unets[0], pim.unets[1]], image_sizes=(16, 32), timesteps=(3, 6), pred_objectives=("noise", "v"),
                     condition_on_text=False).to(device)
    for table in (0, -1):
        pim2.cond_table = table
        out = pim2.sample(batch_size=2, cond_images=cond, start_at_unet_number=2, start_image_or_video=low, device=device, seed=11)
        ref = out if table == 0 else ref
        assert torch.equal(out, ref)


def test_sample_last_returns_what_the_step_computed(device):
    """kd_sample_last (include/kd_engine.h): after one sampler iteration, the UNet output it reports is bit-equal to
    kd_unet_forward on the same x_t / log-SNR, x0-hat follows from it by the library's formula, and the thresholds are
    max(1, quantile_0.95 |x0|) per sample (torch.quantile's interpolation)."""
    import ctypes as C

    from imagen_pytorch import _engine as E
    from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes

    lib = E.load()
    ou = H.oracle_unet("small1", seed=4)
    pu = H.product_unet_like(ou).to(device)
    B, S, T, k = 2, 16, 6, 2
    h = pu.engine(B, S, device, with_text=False)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, 3, S, S, generator=g) * 1.5).to(device)
    x_in = x.clone()
    tables = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=T).step_tables()
    sc = E.kd_schedule_t()
    sc.T = T
    for name, v in tables.items():
        setattr(sc, name, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))
    sa = E.kd_sample_args_t()
    sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times, sa.seed, sa.use_graph = 0, 1, 0.95, 1, 9, 1
    E.check(lib.kd_sample_steps(h, C.byref(sc), C.byref(sa), E.ptr(x), k, k + 1, E.current_stream()))
    pred, x0, thr = torch.empty_like(x), torch.empty_like(x), torch.empty(B, device=device)
    for which, dst in ((0, pred), (1, x0), (2, thr)):
        E.check(lib.kd_sample_last(h, which, E.ptr(dst), E.current_stream()))
    torch.cuda.synchronize()
    ls = tables["log_snr"][k].item()
    fwd = pu(x_in, torch.full((B,), ls, device=device))
    assert torch.equal(pred, fwd)
    alpha, sigma = tables["alpha"][k].item(), tables["sigma"][k].item()
    want_x0 = (x_in - sigma * pred) / max(alpha, 1e-8)
    assert H.rel_l2(x0, want_x0) < 1e-6
    want_thr = torch.quantile(x0.flatten(1).abs(), 0.95, dim=-1).clamp(min=1.0)
    assert torch.equal(thr, want_thr), (thr, want_thr)
    assert lib.kd_sample_last(h, 3, E.ptr(pred), E.current_stream()) != 0 and b"which" in lib.kd_last_error()


def test_return_pil_images_truncates_like_the_library(device):
    """sample(return_pil_images=True) (sample_cond.py:42, sample.py:53): the library maps torchvision's
    ToPILImage over the float images, i.e. mul(255).byte() - truncation, not rounding."""
    import numpy as np

    _, pim = _imagen_pair(device, ["small1"], (16,), (3,), ("noise",))
    ten = pim.sample(batch_size=2, seed=3, device=device)
    pil = pim.sample(batch_size=2, seed=3, device=device, return_pil_images=True)
    assert len(pil) == 2 and pil[0].size == (16, 16) and pil[0].mode == "RGB"
    want = ten.clamp(0, 1).mul(255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(np.stack([np.asarray(p) for p in pil]), want)


def test_plans_of_one_unet_share_their_packed_weights(device):
    """A UNet sampled at several batch / image sizes keeps ONE copy of its packed weights
    (kd_unet_create_shared): the second plan must not grow the weight store, and both plans must
    still match the oracle."""
    from imagen_pytorch import _engine as E

    lib = E.load()
    ou = H.oracle_unet("ultra2", lowres_cond=True, seed=17).eval()
    pu = H.product_unet_like(ou).to(device)
    dv = lambda v: None if v is None else v.to(device)
    sizes = {}
    for B, S in ((2, 32), (3, 32), (1, 64)):
        x, lr, cond, t, tl = _inputs("ultra2", B, S, True, seed=B)
        with torch.no_grad():
            ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
        assert H.rel_l2(got, ref) < FWD_REL_L2
        h = pu.engine(B, S, device, with_text=False)
        sizes[(B, S)] = (lib.kd_unet_weight_bytes(h), lib.kd_unet_hbm_bytes(h))
    w = [v[0] for v in sizes.values()]
    assert w[0] == w[1] == w[2] > 0, f"weight store grew with the number of plans: {sizes}"
    assert len(pu._engines) == 3


def test_engine_mac_count_matches_survey_appendix_b(device):
    """unet2 64->256 (train_ultra_res.py:39-48, 3 cond channels) = 229.2 GMAC/sample."""
    import imagen_pytorch as ip
    from imagen_pytorch import _engine as E

    with torch.device("meta"):
        u = ip.Unet(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                    layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                    init_conv_to_final_conv_residual=True, cond_images_channels=3, lowres_cond=True,
                    cond_on_text=False, text_embed_dim=None)
    u = u.to_empty(device=device)
    h = u.engine(1, 256, device, with_text=False)
    gmac = E.load().kd_unet_macs(h) / 1e9
    assert abs(gmac - 229.2) / 229.2 < 0.01, gmac


def _imagen_pair(device, names, sizes, timesteps, objectives, seed=5):
    import imagen_pytorch as ip

    ous = [H.oracle_unet(n, lowres_cond=i > 0, seed=seed + i) for i, n in enumerate(names)]
    oim = RS.Imagen(ous, image_sizes=sizes, timesteps=timesteps, pred_objectives=objectives, condition_on_text=False)
    pus = [ip.Unet(**{k: v for k, v in u._locals.items()}) for u in oim.unets]
    pim = ip.Imagen(pus, image_sizes=sizes, timesteps=timesteps, pred_objectives=objectives, condition_on_text=False)
    pim.load_state_dict(oim.state_dict(), strict=True)
    return oim, pim.to(device)


@pytest.mark.parametrize("objective", ["noise", "v"])
def test_sample_base_unet_matches_oracle(device, objective):
    """BASELINE config 1 shape (uncond base UNet, DDPM loop) at reduced dim/T: sample_uncond.py:49-55."""
    oim, pim = _imagen_pair(device, ["small1"], (16,), (6,), (objective,))
    nf = RS.generator_noise_fn(77)
    otrace, ptrace = [], []
    ref = oim.sample(noise_fn=nf, batch_size=2, trace=otrace)
    got = pim.sample(noise_fn=nf, batch_size=2, trace=ptrace, device=device)
    assert len(otrace) == len(ptrace) == 6
    for k, (a, b) in enumerate(zip(ptrace, otrace)):
        assert H.rel_l2(a, b) < 1e-4 * (k + 1), (k, H.rel_l2(a, b))
    assert (got.cpu() - ref).abs().max() < SAMPLE_ABS
    # graph replay and eager launches give the same result bit for bit
    got_graph = pim.sample(noise_fn=nf, batch_size=2, device=device, use_graph=True)
    got_eager = pim.sample(noise_fn=nf, batch_size=2, device=device, use_graph=False)
    assert torch.equal(got_graph, got_eager)
    assert (got_graph.cpu() - ref).abs().max() < SAMPLE_ABS


def test_sample_sr_unet_with_cond_and_inpainting_matches_oracle(device):
    """The ultra-res call shape (sample_ultra_res.py:183-195): start at unet 2, cond image, inpaint
    patch + mask with resampling."""
    oim, pim = _imagen_pair(device, ["small1", "small2"], (16, 32), (4, 5), ("noise", "v"))
    g = torch.Generator().manual_seed(9)
    start = torch.rand(1, 3, 16, 16, generator=g)
    cond = torch.rand(1, 3, 64, 64, generator=g)
    inp = torch.rand(1, 3, 32, 32, generator=g)
    mask = torch.zeros(1, 32, 32)
    mask[:, :8, :] = 1
    mask[:, :, :8] = 1
    nf = RS.generator_noise_fn(123)
    kw = dict(batch_size=1, cond_images=cond, start_image_or_video=start, start_at_unet_number=2,
              stop_at_unet_number=2, inpaint_images=inp, inpaint_masks=mask, inpaint_resample_times=3)
    ref = oim.sample(noise_fn=nf, **kw)
    dv = lambda t: t.to(device)
    got = pim.sample(noise_fn=nf, device=device, **{k: (dv(v) if torch.is_tensor(v) else v) for k, v in kw.items()})
    assert got.shape == (1, 3, 32, 32)
    assert (got.cpu() - ref).abs().max() < SAMPLE_ABS
    # known pixels are pasted back exactly (final inpaint step)
    m = mask.bool()[:, None].expand_as(ref)
    assert torch.equal(got.cpu()[m], ref[m])


def test_sample_without_noise_fn_is_seeded_and_bounded(device):
    """Production path: on-device Philox noise; same seed -> same image, different seed -> different."""
    _, pim = _imagen_pair(device, ["small1"], (16,), (5,), ("noise",))
    a = pim.sample(batch_size=2, device=device, seed=42)
    b = pim.sample(batch_size=2, device=device, seed=42)
    c = pim.sample(batch_size=2, device=device, seed=43)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.min() >= 0 and a.max() <= 1
    torch.manual_seed(7)
    d = pim.sample(batch_size=2, device=device)
    torch.manual_seed(7)
    e = pim.sample(batch_size=2, device=device)
    assert torch.equal(d, e)


# ------------------------------------------------------------------------------- engine vs committed golden vectors
import numpy as np  # noqa: E402
from pathlib import Path  # noqa: E402

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("fname,name,lowres", [("unet_small1.npz", "small1", False),
                                               ("unet_small2_lowres.npz", "small2", True)])
def test_engine_unet_forward_matches_golden(device, fname, name, lowres):
    g = np.load(GOLD / fname)
    t = lambda k: torch.from_numpy(g[k]).to(device) if k in g else None
    ou = H.oracle_unet(name, lowres_cond=lowres, seed=int(g["seed"]))
    pu = H.product_unet_like(ou).to(device)
    got = pu(t("x"), t("t"), lowres_cond_img=t("lowres"), lowres_noise_times=t("t_lowres"), cond_images=t("cond"))
    assert H.rel_l2(got, torch.from_numpy(g["y"])) < FWD_REL_L2


def test_engine_cascade_sampler_matches_golden(device):
    import imagen_pytorch as ip

    g = np.load(GOLD / "sampler_cascade.npz")
    seed = int(g["seed"])
    ous = [H.oracle_unet("small1", seed=seed), H.oracle_unet("small2", lowres_cond=True, seed=seed + 1)]
    oim = RS.Imagen(ous, image_sizes=(16, 32), timesteps=(5, 4), pred_objectives=("noise", "v"),
                    condition_on_text=False)
    pim = ip.Imagen([ip.Unet(**u._locals) for u in oim.unets], image_sizes=(16, 32), timesteps=(5, 4),
                    pred_objectives=("noise", "v"), condition_on_text=False)
    pim.load_state_dict(oim.state_dict(), strict=True)
    pim = pim.to(device)
    nf = RS.generator_noise_fn(seed)
    t = lambda k: torch.from_numpy(g[k]).to(device)
    base = pim.sample(noise_fn=nf, batch_size=1, stop_at_unet_number=1, device=device)
    assert (base.cpu() - torch.from_numpy(g["base"])).abs().max() < SAMPLE_ABS
    sr = pim.sample(noise_fn=nf, batch_size=1, cond_images=t("cond"), start_image_or_video=t("base"),
                    start_at_unet_number=2, inpaint_images=t("inpaint"), inpaint_masks=t("mask"),
                    inpaint_resample_times=2, device=device)
    assert (sr.cpu() - torch.from_numpy(g["sr"])).abs().max() < SAMPLE_ABS


def test_patch_grid_on_engine_matches_oracle_driver(device):
    """The ultra-res grid driver (ultra_res/) over the HIP engine vs the same driver over the CPU
    oracle: 2x2 grid, stage-2 only, every patch sampled with the reference's kwargs
    (sample_ultra_res.py:183-195) and inpainted from its finished neighbours."""
    from ultra_res import distributed as D
    from ultra_res import grid as G

    oim, pim = _imagen_pair(device, ["small1", "small2"], (16, 32), (3, 4), ("noise", "v"))
    pos = [(i, j) for i in range(2) for j in range(2)]
    g = torch.Generator().manual_seed(4)
    cond = torch.rand(4, 3, 32, 32, generator=g)
    low = torch.rand(4, 3, 16, 16, generator=g)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 16, 2: 32})
    try:
        def oracle_fn(stage, tasks, lows, conds, ips, ims):
            outs = []
            for t, lo, c, ip, im in zip(tasks, lows, conds, ips, ims):
                nf = RS.generator_noise_fn(1000 + 10 * t[1] + t[2])
                outs.append(oim.sample(noise_fn=nf, batch_size=1, cond_images=c[None], start_image_or_video=lo[None],
                                       start_at_unet_number=stage, stop_at_unet_number=stage, inpaint_images=ip[None],
                                       inpaint_masks=im[None], inpaint_resample_times=2)[0])
            return outs

        def engine_fn(stage, tasks, lows, conds, ips, ims):
            outs = []
            for t, lo, c, ip, im in zip(tasks, lows, conds, ips, ims):
                nf = RS.generator_noise_fn(1000 + 10 * t[1] + t[2])
                dv = lambda v: v[None].to(device)
                outs.append(pim.sample(noise_fn=nf, batch_size=1, cond_images=dv(c), start_image_or_video=dv(lo),
                                       start_at_unet_number=stage, stop_at_unet_number=stage, inpaint_images=dv(ip),
                                       inpaint_masks=dv(im), inpaint_resample_times=2, device=device)[0].cpu())
            return outs

        kw = dict(stages=(2,), patch_pos=[pos], cond_images=[cond], overlap=0.25, num_patches_width=[2],
                  orientations=[-1], lowres=[low])
        ref = D.sample_grids(oracle_fn, **kw)[0]
        got = D.sample_grids(engine_fn, **kw)[0]
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
    for a, b in zip(got, ref):
        assert (a - b).abs().max() < SAMPLE_ABS
    # the overlap strip of patch (1,1) is exactly the bottom strip of patch (0,1)
    assert torch.equal(got[3][:, :8, 8:], got[1][:, -8:, 8:])


def test_patch_grid_with_batched_waves_keeps_the_inpaint_contract(device):
    """imagen_sample_fn(max_batch=4): the patches of a wave share one sample() call (one plan per batch
    size, shared weights).  Noise differs from the one-by-one run, so the check is the contract of the
    grid: every patch's known overlap strip equals its finished neighbour bit for bit, images in [0,1]."""
    from ultra_res import distributed as D
    from ultra_res import grid as G

    _, pim = _imagen_pair(device, ["small1", "small2"], (16, 32), (3, 4), ("noise", "v"))
    n = 3
    pos = [(i, j) for i in range(n) for j in range(n)]
    g = torch.Generator().manual_seed(9)
    cond = torch.rand(n * n, 3, 32, 32, generator=g).to(device)
    low = torch.rand(n * n, 3, 16, 16, generator=g).to(device)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 16, 2: 32})
    calls = []
    orig = pim.sample

    def counting_sample(*a, **k):
        calls.append(k["batch_size"])
        return orig(*a, **k)

    pim.sample = counting_sample
    try:
        fn = D.imagen_sample_fn(lambda stage: pim, 2, device, seed=5, max_batch={2: 4})
        out = D.sample_grids(fn, stages=(2,), patch_pos=[pos], cond_images=[cond], overlap=0.25,
                             num_patches_width=[n], orientations=[-1], lowres=[low], device=device)[0]
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
    assert sorted(calls) == [1, 1, 2, 2, 3], calls      # anti-diagonal waves of a 3x3 grid: 1,2,3,2,1 patches
    idx = {p: k for k, p in enumerate(pos)}
    for (i, j) in pos:
        p = out[idx[(i, j)]]
        assert torch.isfinite(p).all() and p.min() >= 0 and p.max() <= 1
        if i > 0:   # top strip == bottom strip of the patch above
            assert torch.equal(p[:, :8, :], out[idx[(i - 1, j)]][:, -8:, :])
        if j > 0:   # orientation -1: left strip == right strip of the left neighbour
            assert torch.equal(p[:, 8:, :8], out[idx[(i, j - 1)]][:, 8:, -8:])


# ------------------------------------------------------------------------------- text conditioning + guidance (seg-cond path)
SEG_KW = dict(dim=32, dim_mults=(1, 2, 3, 4), cond_dim=64, text_embed_dim=3, num_resnet_blocks=2,
              layer_attns=(False, True, True, True), layer_cross_attns=(False, True, True, True),
              cond_images_channels=4)  # train.py:30-39 at reduced dim


def _seg_pair(device, seed=17, T=4):
    import imagen_pytorch as ip
    from oracle import imagen_ref as R

    ou = H.randomize_(R.Unet(**SEG_KW, cond_on_text=True), seed)
    oim = RS.Imagen([ou], image_sizes=(16,), timesteps=(T,), pred_objectives=("noise",), text_embed_dim=3)
    pim = ip.Imagen([ip.Unet(**oim.unets[0]._locals)], image_sizes=(16,), timesteps=(T,), pred_objectives=("noise",),
                    text_embed_dim=3)
    pim.load_state_dict(oim.state_dict(), strict=True)
    return oim, pim.to(device)


def test_text_conditioned_unet_forward_matches_oracle(device):
    """sample_cond.py:36-48 call shape: text_embeds (B,1,3) = [0.0, 0.5, 0.2], 4 one-hot label planes."""
    oim, pim = _seg_pair(device)
    ou, pu = oim.unets[0], pim.unets[0]
    B, S = 3, 16
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g)
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    text[1, 0] = torch.tensor([0.3, -0.2, 1.0])  # per-sample text
    mask = torch.any(text != 0.0, dim=-1)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, 2 * S, 2 * S), generator=g), 4).permute(0, 3, 1, 2).float()
    dv = lambda v: v.to(device)
    for drop in (0.0, 1.0):
        with torch.no_grad():
            ref = ou(x, t, text_embeds=text, text_mask=mask, cond_images=labels, cond_drop_prob=drop)
        got = pu(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask), cond_images=dv(labels), cond_drop_prob=drop)
        assert H.rel_l2(got, ref) < FWD_REL_L2, (drop, H.rel_l2(got, ref))


@pytest.mark.parametrize("cond_scale", [1.0, 2.5])
def test_text_conditioned_sampling_with_guidance_matches_oracle(device, cond_scale):
    """BASELINE config 2 shape (seg-cond base UNet) at reduced dim/T; cond_scale as sample.py:55-59."""
    oim, pim = _seg_pair(device, T=4)
    B = 2
    g = torch.Generator().manual_seed(3)
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, 16, 16), generator=g), 4).permute(0, 3, 1, 2).float()
    nf = RS.generator_noise_fn(5)
    ref = oim.sample(noise_fn=nf, text_embeds=text, cond_images=labels, cond_scale=cond_scale)
    got = pim.sample(noise_fn=nf, text_embeds=text.to(device), cond_images=labels.to(device), cond_scale=cond_scale,
                     device=device)
    assert (got.cpu() - ref).abs().max() < SAMPLE_ABS


# ------------------------------------------------------------------------------- full-size properties (BASELINE config 3)
def test_full_size_sr_unet_is_deterministic_and_batch_independent(device):
    """At the benchmark's real size (dim 128, 256x256, train_ultra_res.py:39-48) the CPU oracle is too
    slow to run in a test, so the forward is pinned through size-independent properties: repeated
    calls are bit-identical, every sample of a batch equals the same sample run alone (different
    tile mapping, so to fp32 rounding), and the output responds to the conditioning inputs."""
    import imagen_pytorch as ip

    with torch.device("meta"):
        u = ip.Unet(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                    layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                    init_conv_to_final_conv_residual=True, cond_images_channels=3, lowres_cond=True,
                    cond_on_text=False, text_embed_dim=None)
    u = u.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for name, p in u.named_parameters():
            if name.endswith(".g") or name.endswith("norm.weight") or name.endswith("groupnorm.weight") \
                    or name.endswith("norm_cond.weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=device))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g, device=device))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=device) * fan_in ** -0.5)
    S = 256
    x = torch.randn(3, 3, S, S, generator=g, device=device)
    lr = torch.randn(3, 3, S, S, generator=g, device=device)
    cond = torch.rand(3, 3, S, S, generator=g, device=device)
    t = torch.tensor([0.3, -2.0, 4.0], device=device)
    tl = torch.full((3,), -1.1, device=device)
    run = lambda sl: u(x[sl], t[sl], lowres_cond_img=lr[sl], lowres_noise_times=tl[sl], cond_images=cond[sl])
    full = run(slice(0, 3))
    assert torch.isfinite(full).all() and full.std() > 1e-3
    assert torch.equal(full, run(slice(0, 3)))
    for i in range(3):
        single = run(slice(i, i + 1))
        assert H.rel_l2(single, full[i:i + 1]) < 1e-5, i
    other = u(x[:1], t[:1], lowres_cond_img=lr[:1], lowres_noise_times=tl[:1], cond_images=1 - cond[:1])
    assert H.rel_l2(other, full[:1]) > 1e-3
    # the default plan runs the ResnetBlock 3x3 convs as Winograd (F(4x4,3x3) with bf16x3 position GEMMs where the tile
    # slabs fill, the fused F(2x2,3x3) kernel or its batched-GEMM form elsewhere); conv_algo = 1 is the direct implicit GEMM
    # everywhere (a k-ordered fmaf chain): same function
    import ctypes as C
    from imagen_pytorch import _engine as E

    buf = C.create_string_buffer(1 << 20)
    E.check(E.load().kd_unet_profile(u.engine(3, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
    labels = buf.value.decode()
    n_wino = labels.count("wino fused") + labels.count("wino4 gemm") + labels.count("wino gemm")
    assert n_wino >= 40 and labels.count("wino4 gemm bf16x3") >= 10, f"full-size plan lost its Winograd paths: {n_wino}"
    u.conv_algo = 1
    direct = run(slice(0, 3))
    E.check(E.load().kd_unet_profile(u.engine(3, S, device, with_text=False), 1, buf, len(buf), E.current_stream()))
    assert "wino" not in buf.value.decode()
    assert H.rel_l2(direct, full) < FWD_REL_L2


# ------------------------------------------------------------------------------- attention similarity variants
@pytest.mark.parametrize("mode", [1, 2])
def test_qk_norm_attention_variants_match_oracle(device, mode):
    """The library changed its attention similarity between versions and 1.18.5 cannot be inspected
    here, so both alternatives to the default are built and switchable: 1 = `cosine_sim_attn=True` (l2-normalised
    q / k, x16), 2 = learned q_scale / k_scale (x8), selected by the checkpoint's key set.  Self-attention (with
    context and null keys), cross-attention and the text pooling attention against the oracle."""
    import imagen_pytorch as ip
    from oracle import imagen_ref as R

    # (a) ultra-res base UNet: self-attention with context at three levels + cross-attention in the res-blocks
    kw = dict(H.UNET_KW["ultra1"])
    ou = H.randomize_(R.Unet(**kw, cond_on_text=False, text_embed_dim=None, attn_qk_norm=mode), 31).eval()
    plain = {k: v for k, v in ou._locals.items() if k not in ("attn_qk_norm", "cosine_sim_attn")}
    pu = ip.Unet(**plain, cosine_sim_attn=mode == 1)   # the library kwarg gives mode 1; mode 2 comes from the key set
    pu.load_state_dict(ou.state_dict(), strict=True)
    assert pu.attn_qk_norm == mode
    pu = pu.to(device)
    x, lr, cond, t, tl = _inputs("ultra1", 2, 32, False)
    with torch.no_grad():
        ref = ou(x, t, cond_images=cond)
        ref0 = R.Unet(**kw, cond_on_text=False, text_embed_dim=None)
        ref0.load_state_dict({k: v for k, v in ou.state_dict().items() if "_scale" not in k})
        base = ref0.eval()(x, t, cond_images=cond)
    dv = lambda v: None if v is None else v.to(device)
    got = pu(dv(x), dv(t), cond_images=dv(cond))
    assert H.rel_l2(got, ref) < FWD_REL_L2, H.rel_l2(got, ref)
    assert H.rel_l2(ref, base) > 1e-3, "the variant must actually change the function"
    # (b) text-conditioned UNet: the PerceiverResampler attention takes the same switch
    ou = H.randomize_(R.Unet(**SEG_KW, cond_on_text=True, attn_qk_norm=mode), 32).eval()
    pu = ip.Unet(**{k: v for k, v in ou._locals.items() if k != "attn_qk_norm"}, attn_qk_norm=mode)
    pu.load_state_dict(ou.state_dict(), strict=True)
    pu = pu.to(device)
    B, S = 2, 16
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(B, 3, S, S, generator=g), torch.randn(B, generator=g)
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    mask = torch.any(text != 0.0, dim=-1)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, S, S), generator=g), 4).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        ref = ou(x, t, text_embeds=text, text_mask=mask, cond_images=labels)
    got = pu(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask), cond_images=dv(labels))
    assert H.rel_l2(got, ref) < FWD_REL_L2, H.rel_l2(got, ref)


@pytest.mark.parametrize("name,lowres", [("ultra1", False), ("ultra2", True), ("small1", False)])
def test_library_version_forks_match_oracle(device, name, lowres):
    """The two structural forks between library versions: Downsample = Conv2d(4, stride 2, pad 1) and
    mid_attn = residual attention without feed-forward - built by the oracle, loaded into a DEFAULT-constructed
    product Unet (the checkpoint's keys select the fork) and compared on the engine, both forks together and each
    alone; memory-efficient (pre-downsample) and plain (post-downsample + the last level's Parallel) trees."""
    import imagen_pytorch as ip
    from oracle import imagen_ref as R

    kw = dict(H.UNET_KW[name], lowres_cond=lowres, cond_on_text=False, text_embed_dim=None)
    x, lr, cond, t, tl = _inputs(name, 2, 32, lowres)
    dv = lambda v: None if v is None else v.to(device)
    outs = {}
    for forms in (dict(downsample_form="conv4x4", mid_attn_form="residual_attention"), dict(downsample_form="conv4x4"),
                  dict(mid_attn_form="residual_attention"), dict()):
        ou = H.randomize_(R.Unet(**kw, **forms), 77).eval()
        pu = ip.Unet(**kw)
        pu.load_state_dict(ou.state_dict(), strict=True)
        assert (pu.downsample_form, pu.mid_attn_form) == (ou.downsample_form, ou.mid_attn_form)
        pu = pu.to(device)
        with torch.no_grad():
            ref = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
        got = pu(dv(x), dv(t), lowres_cond_img=dv(lr), lowres_noise_times=dv(tl), cond_images=dv(cond))
        err = H.rel_l2(got, ref)
        assert err < FWD_REL_L2, (forms, err)
        outs[tuple(sorted(forms))] = ref
    refs = list(outs.values())
    assert all(H.rel_l2(refs[i], refs[-1]) > 1e-3 for i in range(3)), "a fork must change the function"


def test_patch_grid_overlapped_stage_groups_equal_the_sequential_run(device):
    """sample_grids(overlap_stages=device): inside a generalised wave the stage-2 group runs on the caller's stream
    while the stage-1 group runs from a second host thread on a side stream.  Same tasks, same inputs, same seeds:
    the canvas must equal the sequential run bit for bit (two stages, 3x3 grid, pipelined waves)."""
    from ultra_res import distributed as D
    from ultra_res import grid as G

    _, pim = _imagen_pair(device, ["small2", "small2"], (16, 32), (3, 4), ("noise", "v"))   # both stages take cond images
    n = 3
    pos = [(i, j) for i in range(n) for j in range(n)]
    g = torch.Generator().manual_seed(11)
    cond = torch.rand(n * n, 3, 32, 32, generator=g).to(device)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 16, 2: 32})
    try:
        fn = D.imagen_sample_fn(lambda stage: pim, 2, device, seed=7)
        fn.warm({1: [1], 2: [1]}, cond[0])
        kw = dict(stages=(1, 2), patch_pos=[pos], cond_images=[cond], overlap=0.25, num_patches_width=[n],
                  orientations=[-1], device=device)
        seq = D.sample_grids(fn, **kw)[0]
        par = D.sample_grids(fn, overlap_stages=device, **kw)[0]
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
    assert len(seq) == len(par) == n * n
    for a, b in zip(seq, par):
        assert torch.isfinite(a).all() and torch.equal(a, b)
