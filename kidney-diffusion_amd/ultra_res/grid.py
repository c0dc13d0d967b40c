"""Ultra-res outpainting grid: patch geometry, wavefront order, inpaint-tensor assembly and canvas
stitch — the host-side logic of the reference's patch driver, restated without its queue/process
machinery so that it can be scheduled over `torch.distributed` ranks (see `distributed.py`).

Reference being mirrored (jameshball/kidney-diffusion):
  geometry            sample_ultra_res.py:273-280 (get_patch_width), :304-314 (grid size), :430-431 (canvas)
  dependency rule     sample_ultra_res.py:92-107, :141-143 and get_next_patches :403-412
  orientation         sample_ultra_res.py:423-426
  inpaint patch/mask  sample_ultra_res.py:147-170 (+ fallback crops :128-140)
  tissue filter       sample_ultra_res.py:317-352 (mag 2: HSV threshold, 5x5 erode, 51x51 dilate, any-pixel test)
  cond images         sample_ultra_res.py:356-400 (roll / fill / centre crop)
  stitch              sample_ultra_res.py:434-446, outpainting.py:232-243
Everything here is integer geometry, tensor slicing or pooling; no model arithmetic.  The functions
run on whatever device their tensors live on: with the zoomed image in HBM the conditioning images
and the tissue mask never touch the host (the reference builds them with numpy / skimage / cv2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

PATCH_SIZE = 1024                       # sample_ultra_res.py:31
PATCH_SIZES = {1: 64, 2: 256, 3: 1024}  # sample_ultra_res.py:32
MAG_LEVEL_SIZES = [40000, 6500, 1024]   # ultra_res_patient_dataset.py:18
MAG_LEVEL_SIZES_AIRS = [10000, 3328, 1024]  # ultra_res_airs.py:23

Pos = Tuple[int, int]


def get_patch_width(mag_level: int, sizes: Sequence[int] = MAG_LEVEL_SIZES) -> int:
    """Width of a mag-(level) patch inside the 1024-px image of the level above."""
    return int(sizes[mag_level] * PATCH_SIZE / sizes[mag_level - 1])


@dataclass(frozen=True)
class GridGeometry:
    patch_width: int        # patch footprint in the zoomed (conditioning) image
    patch_dist: int         # stride in the zoomed image
    num_patches_width: int
    out_patch_dist: int     # stride on the output canvas
    canvas_width: int

    @property
    def positions(self) -> List[Pos]:
        n = self.num_patches_width
        return [(i, j) for i in range(n) for j in range(n)]


def grid_geometry(zoomed_width: int, mag_level: int, overlap: float, sizes: Sequence[int] = MAG_LEVEL_SIZES,
                  prefer_in_bounds: bool = False) -> GridGeometry:
    pw = get_patch_width(mag_level, sizes)
    dist = int(pw * (1 - overlap))
    n = 1 + math.ceil((zoomed_width - pw) / dist)
    if prefer_in_bounds:  # the 'airs' variant, sample_ultra_res.py:312-314
        n = max(1, n - 1)
    out_dist = int(PATCH_SIZE * (1 - overlap))
    return GridGeometry(pw, dist, n, out_dist, PATCH_SIZE + (n - 1) * out_dist)


def get_next_patches(patches: Sequence[Pos], orientation: int):
    """Patches none of whose three predecessors are in the list (sample_ultra_res.py:403-412)."""
    s = set(patches)
    ready = [(i, j) for i, j in patches
             if (i - 1, j) not in s and (i, j + orientation) not in s and (i - 1, j + orientation) not in s]
    return ready, [p for p in patches if p not in set(ready)]


def choose_orientation(patch_pos: Sequence[Pos]) -> int:
    """-1 if more patches can start from the top-left than from the top-right (sample_ultra_res.py:423-426)."""
    return -1 if len(get_next_patches(patch_pos, -1)[0]) > len(get_next_patches(patch_pos, 1)[0]) else 1


def dependencies(pos: Pos, patch_pos: Sequence[Pos], orientation: int) -> List[Pos]:
    i, j = pos
    s = set(patch_pos)
    return [p for p in ((i - 1, j), (i, j + orientation), (i - 1, j + orientation)) if p in s]


def wavefronts(patch_pos: Sequence[Pos], orientation: int) -> List[List[Pos]]:
    """Anti-diagonal waves: wave k holds every patch whose predecessors all lie in waves < k.  The
    reference reaches the same partial order by re-queueing patches that are not ready yet."""
    remaining = list(patch_pos)
    waves = []
    while remaining:
        ready, remaining = get_next_patches(remaining, orientation)
        assert ready, "dependency cycle in the patch grid"
        waves.append(ready)
    return waves


STRIP_KINDS = ("above", "next", "corner")   # what a patch takes from (i-1, j), (i, j+o) and (i-1, j+o)


def neighbour_positions(pos: Pos, orientation: int) -> Dict[str, Pos]:
    i, j = pos
    return {"above": (i - 1, j), "next": (i, j + orientation), "corner": (i - 1, j + orientation)}


def cut_strip(kind: str, patch: torch.Tensor, ov: int, orientation: int) -> torch.Tensor:
    """The part of a finished (3,S,S) neighbour that its consumer pastes (sample_ultra_res.py:156-170): the bottom
    `ov` rows of the patch above, the facing `ov` columns of the patch beside it, the facing corner of the diagonal
    one.  These three strips are the whole data dependency between patches."""
    if kind == "above":
        return patch[:, -ov:, :]
    if kind == "next":
        return patch[:, :, -ov:] if orientation == -1 else patch[:, :, :ov]
    if kind == "corner":
        return patch[:, -ov:, -ov:] if orientation == -1 else patch[:, -ov:, :ov]
    raise ValueError(kind)


def strip_shape(kind: str, size: int, ov: int) -> Tuple[int, int, int]:
    return {"above": (3, ov, size), "next": (3, size, ov), "corner": (3, ov, ov)}[kind]


def inpaint_from_strips(strips: Dict[str, Optional[torch.Tensor]], size: int, ov: int, orientation: int):
    """inpaint_patch (3,S,S) and inpaint_mask (S,S) from the three strips (None = no such neighbour), written in the
    reference's order: above, beside, corner (sample_ultra_res.py:156-170)."""
    a, n, an = strips.get("above"), strips.get("next"), strips.get("corner")
    ref = next((t for t in (a, n, an) if t is not None), None)
    kw = dict(device=ref.device, dtype=ref.dtype) if ref is not None else {}
    patch = torch.zeros(3, size, size, **kw)
    mask = torch.zeros(size, size, **kw)
    if a is not None:
        patch[:, :ov, :] = a
        mask[:ov, :] = 1
    if n is not None:
        if orientation == -1:
            patch[:, :, :ov] = n
            mask[:, :ov] = 1
        else:
            patch[:, :, -ov:] = n
            mask[:, -ov:] = 1
    if an is not None:
        if orientation == -1:
            patch[:, :ov, :ov] = an
        else:
            patch[:, :ov, -ov:] = an
    return patch, mask


def fallback_strips(pos: Pos, patch_pos: Sequence[Pos], size: int, overlap: float, orientation: int,
                    num_patches_width: int, cond_image: Optional[torch.Tensor], patch_width: Optional[int]
                    ) -> Dict[str, torch.Tensor]:
    """Strips for neighbours that were filtered out of `patch_pos` but lie inside the image: cut from bilinear-upscaled
    crops of the conditioning image, exactly as sample_ultra_res.py:128-140."""
    i, j = pos
    s = set(patch_pos)
    space_above = i != 0
    space_next = (orientation == 1 and j < num_patches_width - 1) or (orientation == -1 and j > 0)
    space = {"above": space_above, "next": space_next, "corner": space_above and space_next}
    shift = {"above": (-1, 0), "next": (0, orientation), "corner": (-1, orientation)}
    ov = int(overlap * size)
    out = {}
    for kind, p in neighbour_positions(pos, orientation).items():
        if p in s or not space[kind] or cond_image is None:
            continue
        assert patch_width is not None
        dy, dx = shift[kind]
        dist = int(patch_width * (1 - overlap))
        ty = cond_image.shape[1] // 2 - patch_width // 2 + dy * dist
        tx = cond_image.shape[2] // 2 - patch_width // 2 + dx * dist
        crop = cond_image[:3, ty:ty + patch_width, tx:tx + patch_width].unsqueeze(0)
        full = F.interpolate(crop, size=(size, size), mode="bilinear", align_corners=False)[0]
        out[kind] = cut_strip(kind, full, ov, orientation)
    return out


def assemble_inpaint(pos: Pos, patch_pos: Sequence[Pos], done: Dict[Pos, torch.Tensor], size: int, overlap: float,
                     orientation: int, num_patches_width: int, cond_image: Optional[torch.Tensor] = None,
                     patch_width: Optional[int] = None):
    """inpaint_patch (3,S,S) and inpaint_mask (S,S) for one patch from its finished neighbours
    (`done[pos]` = (3,S,S) tensors of the SAME stage).  Neighbours that were filtered out of
    `patch_pos` but lie inside the image fall back to bilinear-upscaled crops of the conditioning
    image, exactly as sample_ultra_res.py:128-140."""
    s = set(patch_pos)
    ov = int(overlap * size)
    strips = fallback_strips(pos, patch_pos, size, overlap, orientation, num_patches_width, cond_image, patch_width)
    for kind, p in neighbour_positions(pos, orientation).items():
        if p in s:
            strips[kind] = cut_strip(kind, done[p], ov, orientation)
    return inpaint_from_strips(strips, size, ov, orientation)


def rgb_to_hsv(img: torch.Tensor) -> torch.Tensor:
    """(3,H,W) RGB in [0,1] -> (3,H,W) HSV with skimage.color.rgb2hsv's conventions (the reference
    calls it at sample_ultra_res.py:321): v = max, s = (max-min)/max (0 where max-min == 0), h from the
    channel that attains the max with blue taking precedence over green over red on ties, h in [0,1)."""
    r, g, b = img[0], img[1], img[2]
    v = img.max(0).values
    delta = v - img.min(0).values
    safe = torch.where(delta == 0, torch.ones_like(delta), delta)
    h = torch.zeros_like(v)
    h = torch.where(r == v, (g - b) / safe, h)
    h = torch.where(g == v, 2.0 + (b - r) / safe, h)
    h = torch.where(b == v, 4.0 + (r - g) / safe, h)
    h = torch.remainder(h / 6.0, 1.0)
    h = torch.where(delta == 0, torch.zeros_like(h), h)
    s = torch.where(delta == 0, torch.zeros_like(v), delta / torch.where(v == 0, torch.ones_like(v), v))
    return torch.stack((h, s, v))


def tissue_mask(zoomed_image: torch.Tensor, version: str = "ultra", erode: int = 5, dilate: int = 51) -> torch.Tensor:
    """(H,W) bool mask of tissue in the mag-1 canvas (sample_ultra_res.py:317-333): HSV threshold
    (hue > 0.5 and saturation > 0.02; value > 0.1 for 'airs'), 5x5 erosion to drop small objects, 51x51
    dilation to grow the mask.  cv2's default border (erode ignores, dilate ignores pixels outside the
    image) is what max-pooling's implicit -inf padding gives.  Runs on the tensor's device."""
    hsv = rgb_to_hsv(zoomed_image[0].float())
    m = (hsv[2] > 0.1) if version == "airs" else ((hsv[0] > 0.5) & (hsv[1] > 0.02))
    m = m.float()[None, None]
    def pool(t, k):   # a k x k max is a k x 1 max of a 1 x k max (2k instead of k^2 reads per pixel: 51 x 51 on 6400^2)
        t = F.max_pool2d(t, (1, k), stride=1, padding=(0, k // 2))
        return F.max_pool2d(t, (k, 1), stride=1, padding=(k // 2, 0))

    m = -pool(-m, erode)
    m = pool(m, dilate)
    return m[0, 0] > 0.5


def tissue_patch_positions(mask: torch.Tensor, geom: GridGeometry) -> List[Pos]:
    """Grid positions whose footprint in the zoomed image touches the mask (sample_ultra_res.py:343-352)."""
    # any() over every footprint at once: a max-pool with the patch as the window, evaluated at the stride
    pw, dist, n = geom.patch_width, geom.patch_dist, geom.num_patches_width
    H, W = mask.shape
    need = (n - 1) * dist + pw
    m = mask.float()
    if need > H or need > W:  # footprints that hang over the edge see only the part inside, as numpy slicing does
        m = F.pad(m, (0, max(0, need - W), 0, max(0, need - H)))
    hit = F.max_pool2d(m[None, None], pw, stride=dist)[0, 0]
    return [(i, j) for i in range(n) for j in range(n) if bool(hit[i, j] > 0.5)]


def center_crop_offset(size: int, crop: int) -> int:
    """First row / column of torchvision's `CenterCrop(crop)` window on an axis of `size` >= `crop` pixels:
    `int(round((size - crop) / 2.0))` (torchvision.transforms.functional.center_crop) - Python's round, i.e.
    half to even: an odd margin such as kidney mag 2's 1024 - 161 = 863 starts at 432, not 431.  The reference
    crops with it at sample_ultra_res.py:391, :393 and :419."""
    return int(round((size - crop) / 2.0))


def center_crop(img: torch.Tensor, crop: int) -> torch.Tensor:
    """`transforms.CenterCrop(crop)(img)` on the last two dims, including torchvision's zero padding of an
    image smaller than the crop (left/top (crop - size) // 2, right/bottom (crop - size + 1) // 2)."""
    h, w = img.shape[-2], img.shape[-1]
    if crop > w or crop > h:
        pl = (crop - w) // 2 if crop > w else 0
        pt = (crop - h) // 2 if crop > h else 0
        pr = (crop - w + 1) // 2 if crop > w else 0
        pb = (crop - h + 1) // 2 if crop > h else 0
        img = F.pad(img, (pl, pr, pt, pb))
        h, w = img.shape[-2], img.shape[-1]
        if crop == w and crop == h:
            return img
    top, left = center_crop_offset(h, crop), center_crop_offset(w, crop)
    return img[..., top:top + crop, left:left + crop]


def cond_images_for_grid(zoomed_image: torch.Tensor, geom: GridGeometry, patch_pos: Sequence[Pos],
                         fill_color: float = 0.95, centre_crop_channels: bool = False) -> torch.Tensor:
    """(N,3|6,1024,1024) conditioning images: the zoomed image shifted so that each patch sits in the
    centre, gaps filled, centre-cropped to 1024 (sample_ultra_res.py:356-400; `v2` adds the
    nearest-upsampled centre patch as 3 extra channels).

    The reference rolls the WHOLE zoomed image per patch (np.roll of 6400 x 6400 x 3 at mag 2) and crops 1024 px
    out of it; only the cropped window is computed here - row / column y of the rolled image is row (y - shift) mod W
    of the source, filled where the reference's slices fill (a shift of 0 fills everything, as its `[shift:]` slice
    does) - which gives the same pixels for 1/39 of the traffic at mag 2."""
    W = zoomed_image.shape[3]
    assert zoomed_image.shape[2] == W, "square zoomed image expected (sample_ultra_res.py:307)"
    src = zoomed_image[0]
    dev = src.device
    out = []
    for i, j in patch_pos:
        cy = i * geom.patch_dist + geom.patch_width // 2
        cx = j * geom.patch_dist + geom.patch_width // 2
        sy, sx = W // 2 - cy, W // 2 - cx
        if W >= PATCH_SIZE:
            top = center_crop_offset(W, PATCH_SIZE)                # sample_ultra_res.py:391
            win = torch.arange(top, top + PATCH_SIZE, device=dev)

            def axis(shift):
                idx = torch.remainder(win - shift, W)
                filled = (win < shift) if shift > 0 else (win >= (W + shift) % W)   # `[shift:]`; shift == 0: every row / column
                return idx, filled

            (ry, fy), (rx, fx) = axis(sy), axis(sx)
            c = src.index_select(1, ry).index_select(2, rx)
            c = torch.where((fy[:, None] | fx[None, :])[None], torch.full_like(c, fill_color), c)
        else:   # narrower than the crop: roll, fill, then torchvision's zero padding
            img = torch.roll(src, shifts=(sy, sx), dims=(1, 2))
            if sy > 0:
                img[:, :sy, :] = fill_color
            else:
                img[:, sy:, :] = fill_color   # sy == 0 fills everything, as the reference's slice does
            if sx > 0:
                img[:, :, :sx] = fill_color
            else:
                img[:, :, sx:] = fill_color
            c = center_crop(img, PATCH_SIZE)
        if centre_crop_channels:
            centre = center_crop(c, geom.patch_width)          # :393
            centre = F.interpolate(centre.unsqueeze(0), PATCH_SIZE, mode="nearest").squeeze(0)
            c = torch.cat((c, centre), 0)
        out.append(c)
    return torch.stack(out)


def bilinear_resize_large(img: torch.Tensor, size: int, band_rows: int = 2048) -> torch.Tensor:
    """`F.interpolate(img, (size, size), mode='bilinear', align_corners=False)` for outputs beyond 2^31 elements (the
    mag-2 canvas: 3 x 40960 x 40960), which torch's device kernel refuses: the same formula (source coordinate
    max(scale (d + 0.5) - 0.5, 0), neighbours i0 and min(i0 + 1, n - 1)) evaluated separably in bands of output rows."""
    _, C, Hs, Ws = img.shape
    dev, dt = img.device, img.dtype

    def taps(n_in, n_out):   # torch's own arithmetic: fp32 scale and source coordinate
        scale = torch.tensor(float(n_in), dtype=torch.float32) / torch.tensor(float(n_out), dtype=torch.float32)
        srcp = ((torch.arange(n_out, dtype=torch.float32) + 0.5) * scale - 0.5).clamp_(min=0).to(dev)
        i0 = srcp.long().clamp_(max=n_in - 1)
        i1 = (i0 + 1).clamp_(max=n_in - 1)
        l1 = (srcp - i0.to(torch.float32)).to(dt)
        return i0, i1, 1 - l1, l1

    y0, y1, ly0, ly1 = taps(Hs, size)
    x0, x1, lx0, lx1 = taps(Ws, size)
    out = torch.empty((1, C, size, size), device=dev, dtype=dt)
    src = img[0]
    for r in range(0, size, band_rows):
        sl = slice(r, min(size, r + band_rows))
        top, bot = src.index_select(1, y0[sl]), src.index_select(1, y1[sl])      # (C, band, Ws)
        tx = lx0 * top.index_select(2, x0) + lx1 * top.index_select(2, x1)
        bx = lx0 * bot.index_select(2, x0) + lx1 * bot.index_select(2, x1)
        out[0, :, sl, :] = ly0[sl][None, :, None] * tx + ly1[sl][None, :, None] * bx
    return out


def stitch_canvas(patches: Sequence[torch.Tensor], patch_pos: Sequence[Pos], geom: GridGeometry,
                  background: Optional[torch.Tensor] = None, patch_size: int = PATCH_SIZE) -> torch.Tensor:
    """Pastes (3,P,P) patches at (i*stride, j*stride) in index order — later patches overwrite the
    overlap, as sample_ultra_res.py:442-446 — onto `background` bilinearly resized to the canvas
    (ultra-res) or zeros (outpainting.py:235)."""
    stride = geom.out_patch_dist  # in output pixels of `patch_size`-wide patches
    width = patch_size + (geom.num_patches_width - 1) * stride
    ref = patches[0]
    if background is not None and 3 * width * width >= 2 ** 31 - 1:
        full = bilinear_resize_large(background.to(ref), width)
    elif background is not None:
        full = F.interpolate(background.to(ref), size=(width, width), mode="bilinear", align_corners=False)
    else:
        full = torch.zeros(1, 3, width, width, device=ref.device, dtype=ref.dtype)
    for idx, (i, j) in enumerate(patch_pos):
        y, x = i * stride, j * stride
        full[0, :, y:y + patch_size, x:x + patch_size] = patches[idx]
    return full
