"""The host drivers of the reference over the HIP engine, at the reference's GEOMETRY (model width reduced to dim 32
so that the CPU oracle can run the same driver):

  * one mag-2 level at its real numbers (sample_ultra_res.py:304-352, 414-448): a 6400 x 6400 mag-1 canvas, patch
    width 161 / stride 120 / 53 x 53 candidates, the tissue filter (HSV threshold, 5x5 erode, 51x51 dilate), fallback
    crops of the conditioning image for the filtered-out neighbours (:128-140), the 40960 x 40960 canvas - the grid
    driver over the engine against THE SAME driver over the oracle, patch by patch;
  * `outpaint_canvas` 3 x 3 (outpainting.py:173-243) over the engine against the same driver over the oracle;
  * the chain mag 0 -> mag 1 -> mag 2 (sample_ultra_res.py:463-469) on the engine with every image in HBM.
"""
import time

import pytest
import torch

import helpers as H
from oracle import imagen_ref as R
from oracle import sampler_ref as RS

pytestmark = pytest.mark.gpu

SAMPLE_ABS = 2e-3
IMAGEN_KW = dict(image_sizes=(64, 256, 1024), pred_objectives=("noise", "noise", "noise"), condition_on_text=False)


def _pair(device, cond_channels, timesteps, seed):
    """Oracle Imagen (CPU) and product Imagen (device) with the same weights: the three UNets of
    train_ultra_res.py:29-60 at dim 32."""
    import imagen_pytorch as ip

    ous = []
    for s, name in ((1, "ultra1"), (2, "ultra2"), (3, "ultra3")):
        kw = dict(H.UNET_KW[name], cond_images_channels=cond_channels)
        ous.append(H.fast_oracle(H.randomize_(R.Unet(**kw, lowres_cond=s > 1, cond_on_text=False, text_embed_dim=None), seed + s).eval()))
    oim = RS.Imagen(ous, timesteps=timesteps, **IMAGEN_KW)
    pim = ip.Imagen([ip.Unet(**u._locals) for u in oim.unets], timesteps=timesteps, random_crop_sizes=(None, None, 256),
                    **IMAGEN_KW)
    pim.load_state_dict(oim.state_dict(), strict=True)
    return oim, pim.to(device)


def _noise_fn(stage, task):
    return RS.generator_noise_fn(7000 + 100003 * stage + 1009 * task[1] + task[2])


def _fns(oim, pim, device, resample):
    def call(im, stage, t, lo, c, ip_, im_, dev):
        dv = lambda v: None if v is None else v[None].to(dev)
        kw = dict(noise_fn=_noise_fn(stage, t), batch_size=1, cond_images=dv(c), start_image_or_video=dv(lo),
                  start_at_unet_number=stage, stop_at_unet_number=stage)
        if ip_ is not None:
            kw.update(inpaint_images=dv(ip_), inpaint_masks=dv(im_), inpaint_resample_times=resample)
        if dev.type == "cuda":
            kw["device"] = dev
        return im.sample(**kw)[0]

    def oracle_fn(stage, tasks, lows, conds, ips, ims):
        return [call(oim, stage, *a, torch.device("cpu")) for a in zip(tasks, lows, conds, ips, ims)]

    def engine_fn(stage, tasks, lows, conds, ips, ims):
        return [call(pim, stage, *a, device) for a in zip(tasks, lows, conds, ips, ims)]

    return oracle_fn, engine_fn


def _mag1_canvas_with_tissue():
    """A stand-in for a mag-1 canvas: near-white background, one purple blob (hue 0.8) that the filter keeps, a
    3 x 3 speck that the 5 x 5 erosion removes and a grey block (no hue)."""
    g = torch.Generator().manual_seed(31)
    img = torch.rand(1, 3, 6400, 6400, generator=g) * 0.01 + 0.95      # saturation <= 0.0105: below the 0.02 threshold
    img[0, :, 3010:3050, 2900:3140] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]
    img[0, :, 3050:3200, 2900:2960] = torch.tensor([0.70, 0.30, 0.8])[:, None, None]   # an L: the patch set is not a rectangle
    img[0, :, 900:903, 5100:5103] = torch.tensor([0.7, 0.3, 0.8])[:, None, None]
    img[0, :, 5000:5200, 1000:1300] = 0.5
    return img


def test_mag2_level_at_the_reference_geometry_matches_the_driver_over_the_oracle(device):
    from ultra_res import distributed as D
    from ultra_res import grid as G
    from ultra_res import pipeline as P

    zoomed = _mag1_canvas_with_tissue()
    geom, pos = P.level_patches(zoomed, 2, 0.25)
    assert (geom.patch_width, geom.patch_dist, geom.num_patches_width, geom.canvas_width) == (161, 120, 53, 40960)
    geom_d, pos_d = P.level_patches(zoomed.to(device), 2, 0.25)
    assert pos_d == pos and geom_d == geom            # the tissue filter on the device picks the same patches
    rows, cols = sorted({p[0] for p in pos}), sorted({p[1] for p in pos})
    assert 4 <= len(pos) <= 12 and len(pos) < len(rows) * len(cols), pos   # holes inside the bounding box
    assert all(0 < p[0] < 52 and 0 < p[1] < 52 for p in pos)   # every border patch of the set needs fallback crops
    o = G.choose_orientation(pos)
    oim, pim = _pair(device, 3, (2, 2, 1), seed=400)
    oracle_fn, engine_fn = _fns(oim, pim, device, resample=1)
    cond = G.cond_images_for_grid(zoomed, geom, pos)
    cond_d = G.cond_images_for_grid(zoomed.to(device), geom, pos)
    assert torch.equal(cond, cond_d.cpu())
    kw = dict(stages=(1, 2, 3), patch_pos=[pos], overlap=0.25, num_patches_width=[53], orientations=[o],
              patch_width=geom.patch_width)
    got = D.sample_grids(engine_fn, cond_images=[cond_d], device=device, **kw)[0]
    t0 = time.perf_counter()
    ref = D.sample_grids(oracle_fn, cond_images=[cond], **kw)[0]
    print(f"mag 2: {len(pos)} of 2809 candidate patches {pos}, orientation {o}; oracle driver {time.perf_counter() - t0:.0f} s")
    idx = {p: k for k, p in enumerate(pos)}
    worst = 0.0
    for p in pos:
        a, b = got[idx[p]].cpu(), ref[idx[p]]
        worst = max(worst, float((a - b).abs().max()))
        # the known strips: a finished neighbour's strip where it was generated, else the bilinear crop of the
        # conditioning image (:128-140) - pasted bit for bit after the last step on both paths
        if (p[0] - 1, p[1]) in idx:
            assert torch.equal(a[:, :256, :], got[idx[(p[0] - 1, p[1])]].cpu()[:, -256:, :])
        else:
            dist_ = geom.patch_dist
            ty = 512 - geom.patch_width // 2 - dist_
            tx = 512 - geom.patch_width // 2
            crop = cond[idx[p]][:3, ty:ty + 161, tx:tx + 161][None]
            up = torch.nn.functional.interpolate(crop, size=(1024, 1024), mode="bilinear", align_corners=False)[0]
            cols_free = slice(256, None) if o == -1 else slice(None, -256)   # the side strip is written after the top one
            assert torch.allclose(b[:, :256, cols_free], up[:, -256:, cols_free], atol=1e-6)
            assert torch.allclose(a[:, :256, cols_free], up[:, -256:, cols_free], atol=1e-5)
    print(f"mag 2 patches, engine driver vs oracle driver: worst max|diff| {worst:.3e}")
    assert worst < 3 * SAMPLE_ABS, worst
    # the level as the product runs it: everything in HBM, canvas 40960 x 40960 (20 GB), patches pasted in index order
    canvas = P.generate_high_res_image(engine_fn, zoomed.to(device), 2, overlap=0.25, device=device)
    assert canvas.is_cuda and tuple(canvas.shape) == (1, 3, 40960, 40960)
    for k, (i, j) in enumerate(pos):   # a patch's top-left 768 x 768 is never overwritten by a later patch
        assert torch.equal(canvas[0, :, i * 768:i * 768 + 768, j * 768:j * 768 + 768], got[k][:, :768, :768]), (i, j)
    # the background: the mag-1 canvas enlarged bilinearly (align_corners False), checked at random points in fp64
    gen = torch.Generator().manual_seed(5)
    ys = torch.randint(0, 18000, (2000,), generator=gen)
    xs = torch.randint(0, 40960, (2000,), generator=gen)
    z = zoomed[0].double()

    def tap(d):
        s = ((d.double() + 0.5) * (6400 / 40960) - 0.5).clamp(min=0)
        i0 = s.floor().long().clamp(max=6399)
        return i0, (i0 + 1).clamp(max=6399), s - i0

    y0, y1, ly = tap(ys)
    x0, x1, lx = tap(xs)
    want = (1 - ly) * ((1 - lx) * z[:, y0, x0] + lx * z[:, y0, x1]) + ly * ((1 - lx) * z[:, y1, x0] + lx * z[:, y1, x1])
    have = canvas[0][:, ys.to(device), xs.to(device)].cpu().double()
    assert float((have - want).abs().max()) < 2e-4   # fp32 source coordinates at 40960 px, as torch computes them
    del canvas


def test_outpaint_canvas_3x3_matches_the_driver_over_the_oracle(device):
    """outpainting.py:173-243: unconditional 3 x 3 grid, orientation -1, zero canvas of width 1024 + 2 * 768."""
    from ultra_res import distributed as D

    oim, pim = _pair(device, 0, (2, 2, 1), seed=500)
    oracle_fn, engine_fn = _fns(oim, pim, device, resample=1)
    got = D.outpaint_canvas(engine_fn, 3, overlap=0.25, device=device)[0]
    t0 = time.perf_counter()
    ref = D.outpaint_canvas(oracle_fn, 3, overlap=0.25)[0]
    print(f"outpaint 3x3: oracle driver {time.perf_counter() - t0:.0f} s")
    assert tuple(got.shape) == tuple(ref.shape) == (1, 3, 2560, 2560)
    err = float((got.cpu() - ref).abs().max())
    print(f"outpaint 3x3 canvas, engine driver vs oracle driver: max|diff| {err:.3e}")
    assert err < 3 * SAMPLE_ABS, err
    assert got.min() >= 0 and got.max() <= 1


def test_three_level_chain_mag0_mag1_mag2_on_the_engine(device):
    """sample_ultra_res.py:463-469 with every image in HBM: mag 0 (one unconditional cascade sample) -> mag 1 (the full
    8 x 8 grid over it, canvas 6400^2) -> mag 2 (tissue-filtered patches of the mag-1 canvas, at most six here, canvas
    40960^2).  Properties: sizes, value range of everything generated, reproducibility, the mag-1 canvas holding the
    mag-0 image enlarged wherever nothing is pasted... which at mag 1 is nowhere (the grid covers it)."""
    from ultra_res import distributed as D
    from ultra_res import grid as G
    from ultra_res import pipeline as P

    _, pim0 = _pair(device, 0, (2, 2, 1), seed=600)
    _, pim1 = _pair(device, 3, (1, 1, 1), seed=610)
    fns = {0: D.imagen_sample_fn(lambda st: pim0, 1, device, seed=21),
           1: D.imagen_sample_fn(lambda st: pim1, 1, device, seed=22),
           2: D.imagen_sample_fn(lambda st: pim1, 2, device, seed=23)}
    picked = {}

    def keep(level, pos):
        if level == 2:   # whatever the tissue filter finds on this random canvas, thinned to six; two fixed ones if nothing
            pos = pos[len(pos) // 2:len(pos) // 2 + 6] if pos else [(26, 26), (26, 27)]
        picked[level] = pos
        return pos

    mag0, mag1, mag2 = P.generate_all_levels(fns, overlap=0.25, device=device, patch_filter=keep)
    assert mag0.is_cuda and mag1.is_cuda and mag2.is_cuda
    assert tuple(mag0.shape) == (1, 3, 1024, 1024) and tuple(mag1.shape) == (1, 3, 6400, 6400)
    assert tuple(mag2.shape) == (1, 3, 40960, 40960)
    assert len(picked[1]) == 64 and 1 <= len(picked[2]) <= 6
    for img in (mag0, mag1):
        assert torch.isfinite(img).all() and img.min() >= 0 and img.max() <= 1
    i, j = picked[2][-1]
    win = mag2[0, :, i * 768:i * 768 + 1024, j * 768:j * 768 + 1024]
    assert torch.isfinite(win).all() and win.min() >= 0 and win.max() <= 1 and win.std() > 0
    del mag2
    again = P.generate_all_levels(fns, overlap=0.25, device=device, patch_filter=lambda lv, p: p if lv == 1 else picked[2][:1])
    assert torch.equal(again[0], mag0) and torch.equal(again[1], mag1)   # seeded per patch: reproducible
