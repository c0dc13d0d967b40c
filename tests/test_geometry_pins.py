"""Host geometry of the ultra-res driver pinned to tests/golden/geometry.json, which
tests/golden/make_geometry_golden.py derives from the formulas of the third-party calls the reference makes
(torchvision's CenterCrop) and from the reference's own integer arithmetic - not from this repo's product
code.  The checks below read WHICH pixels a product function selected through coordinate-coded images, so
they share no slicing code with ultra_res/grid.py either."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from ultra_res import grid as G
from ultra_res import pipeline as P

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "geometry.json").read_text())


def _coord_image(h, w):
    """(3,h,w): channel 0 = row index, channel 1 = column index, channel 2 = 1 (0 marks padding)."""
    r = torch.arange(h, dtype=torch.float32)[:, None].expand(h, w)
    c = torch.arange(w, dtype=torch.float32)[None, :].expand(h, w)
    return torch.stack((r, c, torch.ones(h, w)))


@pytest.mark.parametrize("case", GOLD["center_crop_offsets"], ids=lambda c: f"{c['size']}to{c['crop']}")
def test_center_crop_matches_torchvision_window(case):
    size, crop = case["size"], case["crop"]
    pl, pt, pr, pb, top, left = case["window"]
    out = G.center_crop(_coord_image(size, size), crop)
    assert tuple(out.shape) == (3, crop, crop)
    # expected: pad with zeros, then take [top:top+crop, left:left+crop]
    exp = np.zeros((3, size + pt + pb, size + pl + pr), dtype=np.float32)
    exp[:, pt:pt + size, pl:pl + size] = _coord_image(size, size).numpy()
    exp = exp[:, top:top + crop, left:left + crop]
    assert np.array_equal(out.numpy(), exp)
    if size >= crop:
        assert G.center_crop_offset(size, crop) == top


def test_kidney_mag2_centre_crop_starts_at_432():
    """1024 - 161 = 863 is odd: torchvision rounds 431.5 half-to-even to 432 (sample_ultra_res.py:393, :419)."""
    assert GOLD["kidney_mag2"]["patch_width"] == 161
    assert GOLD["kidney_mag2"]["crop_to_patch_width"][4:] == [432, 432]
    assert G.center_crop_offset(1024, 161) == 432
    assert G.center_crop_offset(1024, 166) == 429


@pytest.mark.parametrize("name,mag,sizes,airs", [("kidney_mag1", 1, G.MAG_LEVEL_SIZES, False),
                                                 ("kidney_mag2", 2, G.MAG_LEVEL_SIZES, False),
                                                 ("airs_mag1", 1, G.MAG_LEVEL_SIZES_AIRS, True)])
def test_grid_geometry_matches_reference_arithmetic(name, mag, sizes, airs):
    g = GOLD[name]
    geom = G.grid_geometry(g["zoomed_width"], mag, 0.25, sizes=sizes, prefer_in_bounds=airs)
    assert (geom.patch_width, geom.patch_dist, geom.num_patches_width, geom.out_patch_dist, geom.canvas_width) == \
        (g["patch_width"], g["patch_dist"], g["num_patches_width"], g["out_patch_dist"], g["canvas_width"])


def _expected_cond(zoomed, shift, crop_top, fill):
    """Restatement of sample_ultra_res.py:372-391 by index arithmetic on numpy arrays: roll, fill, crop."""
    _, W, _ = zoomed.shape
    sy, sx = shift
    rows = (np.arange(W) - sy) % W
    cols = (np.arange(W) - sx) % W
    img = zoomed[:, rows][:, :, cols].copy()
    fr = np.zeros(W, bool)
    fc = np.zeros(W, bool)
    if sy > 0:
        fr[:sy] = True
    else:
        fr[sy:] = True      # python slice semantics of the reference: shift 0 fills everything
    if sx > 0:
        fc[:sx] = True
    else:
        fc[sx:] = True
    img[:, fr, :] = fill
    img[:, :, fc] = fill
    return img[:, crop_top:crop_top + 1024, crop_top:crop_top + 1024]


def test_cond_images_shift_fill_and_crops_at_kidney_mag2_numbers():
    """cond_images_for_grid with the kidney mag-2 patch width (161) and stride (120) on a 2048-wide stand-in
    of the 6400-wide mag-1 canvas: the CenterCrop(1024) window, the shift, the fill and - for version v2 - the
    CenterCrop(161) window (offset 432) of every pinned patch."""
    g = GOLD["kidney_mag2"]
    W = 2048
    zoomed = _coord_image(W, W)
    geom = G.GridGeometry(g["patch_width"], g["patch_dist"], 8, g["out_patch_dist"], 0)
    pos = [(0, 0), (1, 2), (7, 7), (4, 2)]
    out = G.cond_images_for_grid(zoomed[None], geom, pos, fill_color=0.95, centre_crop_channels=True)
    assert tuple(out.shape) == (4, 6, 1024, 1024)
    top1 = int(round((W - 1024) / 2.0))
    o2 = g["crop_to_patch_width"][4]
    for n, (i, j) in enumerate(pos):
        cy, cx = i * g["patch_dist"] + g["patch_width"] // 2, j * g["patch_dist"] + g["patch_width"] // 2
        exp = _expected_cond(zoomed.numpy(), (W // 2 - cy, W // 2 - cx), top1, np.float32(0.95))
        assert np.array_equal(out[n, :3].numpy(), exp), (i, j)
        centre = exp[:, o2:o2 + 161, o2:o2 + 161]
        src = (np.arange(1024) * (161 / 1024)).astype(np.int64)   # F.interpolate(mode='nearest'): floor(dst * scale)
        assert np.array_equal(out[n, 3:].numpy(), centre[:, src][:, :, src]), (i, j)
        # the patch's own footprint sits at the centre of the conditioning image
        assert out[n, 0, 512, 512] == cy and out[n, 1, 512, 512] == cx


def test_ignore_unet_1_start_images_use_the_torchvision_window():
    """generate_high_res_image(ignore_unet_1=True): lowres = CenterCrop(patch_width)(cond_image)
    (sample_ultra_res.py:417-420) - offset 432 at the kidney mag-2 patch width."""
    seen = {}

    def fake_sample(stage, tasks, lows, conds, ips, ims):
        seen.setdefault(stage, (lows, conds))
        S = G.PATCH_SIZES[stage]
        return [torch.zeros(3, S, S) for _ in tasks]

    zoomed = _coord_image(1024, 1024)[None]
    real = G.get_patch_width
    G.get_patch_width = lambda mag, sizes=G.MAG_LEVEL_SIZES: 161   # mag-2 footprint on a 1024-wide stand-in
    try:
        P.generate_high_res_image(fake_sample, zoomed, 1, overlap=0.25, ignore_unet_1=True, patch_pos=[(0, 0)])
    finally:
        G.get_patch_width = real
    assert 1 not in seen and 2 in seen
    low, cond = seen[2][0][0], seen[2][1][0]
    assert tuple(low.shape) == (3, 161, 161)
    assert torch.equal(low, cond[:3, 432:432 + 161, 432:432 + 161])
