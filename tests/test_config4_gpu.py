"""BASELINE configs[4] at the reference's dims: the ultra-res outpainting grid of 1024-px patches through the
three-stage cascade 64 -> 256 -> 1024 (train_ultra_res.py:27-92), every patch sampled with the kwargs of
sample_ultra_res.py:183-195 and inpainted from its finished neighbours (:92-174).

  * 1x2 grid, stages 1 -> 2 -> 3, inpaint_resample 2, timesteps (2, 2, 1): the grid driver over the HIP
    engine against THE SAME driver over the CPU oracle (identical weights, injected noise keyed per patch);
  * 8x8 grid (64 patches, canvas 6400 x 6400), one timestep per stage: too large for the oracle, checked
    through the properties the path guarantees: every patch's known overlap strips equal its finished
    neighbours bit for bit in all three stages' final output, the stitched canvas has the reference's size and holds each
    patch where sample_ultra_res.py:442-446 pastes it, a second run is bit-identical (seeded), values in [0, 1].
"""
import time

import pytest
import torch

import helpers as H
from oracle import imagen_ref as R
from oracle import sampler_ref as RS

pytestmark = pytest.mark.gpu

SAMPLE_ABS = 2e-3
F_, T_ = False, True
ULTRA_KW = {   # train_ultra_res.py:29-60 (magnification level > 0: three conditioning channels)
    1: dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
            layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=3),
    2: dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True, layer_attns=(F_, F_, F_, T_),
            layer_cross_attns=(F_, F_, T_, T_), init_conv_to_final_conv_residual=True, cond_images_channels=3),
    3: dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True, layer_attns=False,
            layer_cross_attns=(F_, F_, F_, T_), init_conv_to_final_conv_residual=True, cond_images_channels=3),
}
IMAGEN_KW = dict(image_sizes=(64, 256, 1024), pred_objectives=("noise", "noise", "noise"), condition_on_text=False)


@pytest.fixture(scope="module")
def cascade(device):
    """The three UNets of train_ultra_res.py at full dims, once per module: oracle modules (CPU) and the product
    Imagen (device) holding the same weights."""
    import imagen_pytorch as ip

    ous = [H.fast_oracle(H.randomize_(R.Unet(**ULTRA_KW[s], lowres_cond=s > 1, cond_on_text=False, text_embed_dim=None), 300 + s).eval())
           for s in (1, 2, 3)]

    def make(T):
        T = (T, T, T) if isinstance(T, int) else tuple(T)
        oim = RS.Imagen(ous, timesteps=T, **IMAGEN_KW)
        pim = ip.Imagen([ip.Unet(**u._locals) for u in oim.unets], timesteps=T, random_crop_sizes=(None, None, 256),
                        **IMAGEN_KW)
        pim.load_state_dict(oim.state_dict(), strict=True)
        return oim, pim.to(device)

    return make


def _noise_fn(task):
    return RS.generator_noise_fn(5000 + 100 * task[0] + 10 * task[1] + task[2])


def test_c5_two_patch_grid_three_stages_full_dims_matches_oracle_driver(device, cascade):
    from ultra_res import distributed as D
    from ultra_res import grid as G

    R_TIMES = 2
    # two timesteps in stages 1 and 2 (first- and last-step branches of the loop), ONE in stage 3, where a forward is
    # 12 TFLOP on the host: with the two resampling iterations per timestep that is still two 1024-px forwards per patch
    # (the two-timestep loop of the stage-3 UNet at full dims runs in test_fullsize_gpu's cascade test)
    oim, pim = cascade((2, 2, 1))
    pos = [(0, 0), (0, 1)]
    geom = G.grid_geometry(1024, 1, 0.25)
    g = torch.Generator().manual_seed(77)
    zoomed = torch.rand(1, 3, 1024, 1024, generator=g)
    cond = G.cond_images_for_grid(zoomed, geom, pos)   # (2, 3, 1024, 1024), as get_cond_images builds them
    assert tuple(cond.shape) == (2, 3, 1024, 1024)

    def oracle_fn(stage, tasks, lows, conds, ips, ims):
        outs = []
        for t, lo, c, ip_, im in zip(tasks, lows, conds, ips, ims):
            outs.append(oim.sample(noise_fn=_noise_fn(t), batch_size=1, cond_images=c[None],
                                   start_image_or_video=None if lo is None else lo[None], start_at_unet_number=stage,
                                   stop_at_unet_number=stage, inpaint_images=ip_[None], inpaint_masks=im[None],
                                   inpaint_resample_times=R_TIMES)[0])
        return outs

    def engine_fn(stage, tasks, lows, conds, ips, ims):
        outs = []
        dv = lambda v: None if v is None else v[None].to(device)
        for t, lo, c, ip_, im in zip(tasks, lows, conds, ips, ims):
            outs.append(pim.sample(noise_fn=_noise_fn(t), batch_size=1, cond_images=dv(c), start_image_or_video=dv(lo),
                                   start_at_unet_number=stage, stop_at_unet_number=stage, inpaint_images=dv(ip_),
                                   inpaint_masks=dv(im), inpaint_resample_times=R_TIMES, device=device)[0].cpu())
        return outs

    kw = dict(stages=(1, 2, 3), patch_pos=[pos], cond_images=[cond], overlap=0.25, num_patches_width=[8],
              orientations=[-1], patch_width=geom.patch_width)
    got = D.sample_grids(engine_fn, **kw)[0]
    t0 = time.perf_counter()
    ref = D.sample_grids(oracle_fn, **kw)[0]
    print(f"oracle driver, 2 patches x 3 stages x T=(2, 2, 1) x R=2 at full dims: {time.perf_counter() - t0:.0f} s on the host")
    for n, (a, b) in enumerate(zip(got, ref)):
        assert tuple(a.shape) == (3, 1024, 1024)
        err = float((a - b).abs().max())
        print(f"configs[4] patch {pos[n]}: max|diff| {err:.3e}")
        assert err < 3 * SAMPLE_ABS, (n, err)   # three chained stages, each within SAMPLE_ABS of its own inputs
    # patch (0, 1) was inpainted from patch (0, 0): its left 256 columns are (0, 0)'s right 256 columns, bit for bit
    assert torch.equal(got[1][:, :, :256], got[0][:, :, -256:])
    assert torch.equal(ref[1][:, :, :256], ref[0][:, :, -256:])


def test_c5_full_8x8_grid_full_dims_properties(device, cascade):
    from ultra_res import distributed as D
    from ultra_res import grid as G

    _, pim = cascade(1)
    geom = G.grid_geometry(1024, 1, 0.25)
    assert geom.num_patches_width == 8 and geom.canvas_width == 6400 and geom.out_patch_dist == 768
    n = 8
    pos = [(i, j) for i in range(n) for j in range(n)]
    g = torch.Generator().manual_seed(78)
    zoomed = torch.rand(1, 3, 1024, 1024, generator=g).to(device)
    cond = G.cond_images_for_grid(zoomed, geom, pos)
    models = {}

    def load(stage):   # the Imagen is shared by the three stages here (all three UNets are resident)
        return models.setdefault(stage, pim)

    def run():
        fn = D.imagen_sample_fn(load, 1, device, use_graph=True, seed=4321)
        return D.sample_grids(fn, (1, 2, 3), [pos], [cond], 0.25, [n], patch_width=geom.patch_width, device=device)[0]

    a = run()
    assert len(a) == 64 and all(tuple(p.shape) == (3, 1024, 1024) for p in a)
    assert all(torch.isfinite(p).all() and p.min() >= 0 and p.max() <= 1 for p in a)
    o = G.choose_orientation(pos)
    idx = {p: k for k, p in enumerate(pos)}
    w = 256   # int(0.25 * 1024) known columns / rows (sample_ultra_res.py:147-170)
    checked = 0
    for (i, j) in pos:
        me = a[idx[(i, j)]]
        if i > 0:   # above: my top strip is its bottom strip
            assert torch.equal(me[:, :w, :], a[idx[(i - 1, j)]][:, -w:, :]), (i, j, "above")
            checked += 1
        jn = j + o
        if 0 <= jn < n:   # next_to: the shared vertical strip
            nb = a[idx[(i, jn)]]
            if o == -1:
                assert torch.equal(me[:, w if i > 0 else 0:, :w], nb[:, w if i > 0 else 0:, -w:]), (i, j, "next_to")
            else:
                assert torch.equal(me[:, w if i > 0 else 0:, -w:], nb[:, w if i > 0 else 0:, :w]), (i, j, "next_to")
            checked += 1
    assert checked == 2 * n * (n - 1)
    canvas = G.stitch_canvas(a, pos, geom, background=zoomed)
    assert tuple(canvas.shape) == (1, 3, 6400, 6400)
    for (i, j) in ((0, 0), (3, 4), (7, 7)):   # later patches overwrite the overlaps: the last writer of a region wins
        y, x = i * 768, j * 768
        y1 = 1024 if i == n - 1 else 768
        x1 = 1024 if j == n - 1 else 768
        assert torch.equal(canvas[0, :, y:y + y1, x:x + x1], a[idx[(i, j)]][:, :y1, :x1])
    b = run()
    assert all(torch.equal(p, q) for p, q in zip(a, b)), "seeded grid run is not reproducible"
