"""One magnification level of the ultra-res pipeline — `generate_high_res_image` of the reference
(sample_ultra_res.py:414-448) over the wavefront scheduler: conditioning images and patch positions
from the zoomed image of the level above (tissue filter at mag 2), orientation choice, the three
stages over the (filtered) grid, paste onto the bilinearly enlarged zoomed image.

The reference chains it as mag 0 (one unconditional 64->256->1024 sample) -> mag 1 (8x8 grid, canvas
6400^2) -> mag 2 (53x53 candidates filtered by the tissue mask, canvas 40960^2), reloading a model per
stage and level (:264-270, :451-494); here the caller supplies one `sample_fn` per level (e.g.
`distributed.imagen_sample_fn`, which keeps the three stages of a level resident).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import distributed as D
from . import grid as G


def level_patches(zoomed_image: torch.Tensor, mag_level: int, overlap: float, version: str = "ultra"
                  ) -> Tuple[G.GridGeometry, List[G.Pos]]:
    """Grid geometry and the patch positions to generate at `mag_level` (sample_ultra_res.py:304-352):
    every position at mag 1, only those whose footprint touches tissue at mag 2."""
    airs = version == "airs"
    geom = G.grid_geometry(zoomed_image.shape[3], mag_level, overlap,
                           sizes=G.MAG_LEVEL_SIZES_AIRS if airs else G.MAG_LEVEL_SIZES, prefer_in_bounds=airs)
    if mag_level == 2:
        return geom, G.tissue_patch_positions(G.tissue_mask(zoomed_image, version=version), geom)
    return geom, geom.positions


def generate_high_res_image(sample_fn: Callable, zoomed_image: torch.Tensor, mag_level: int, overlap: float = 0.25,
                            version: str = "ultra", ignore_unet_1: bool = False, fill_color: Optional[float] = None,
                            group=None, device: Optional[torch.device] = None,
                            patch_pos: Optional[Sequence[G.Pos]] = None) -> torch.Tensor:
    """(1,3,W,W) canvas of `mag_level` conditioned on `zoomed_image` (1,3,w,w) of the level above.
    `version` 'v2' adds the nearest-upsampled centre crop as 3 more conditioning channels (:391-397);
    `ignore_unet_1` starts at stage 2 from the centre crop of each conditioning image (:417-420)."""
    geom, pos = level_patches(zoomed_image, mag_level, overlap, version)
    if patch_pos is not None:
        pos = list(patch_pos)
    if not pos:  # nothing to generate (a mag-2 canvas without tissue): the enlarged background alone
        return torch.nn.functional.interpolate(zoomed_image, size=(geom.canvas_width, geom.canvas_width),
                                               mode="bilinear", align_corners=False)
    if fill_color is None:
        fill_color = 0.0 if version == "airs" else 0.95  # sample_ultra_res.py:374-377
    cond = G.cond_images_for_grid(zoomed_image, geom, pos, fill_color=fill_color, centre_crop_channels=version == "v2")
    lowres, stages = None, (1, 2, 3)
    if ignore_unet_1:
        lowres = [G.center_crop(cond[:, :3], geom.patch_width)]   # transforms.CenterCrop(patch_width), :419
        stages = (2, 3)
    out = D.sample_grids(sample_fn, stages, [pos], [cond], overlap, [geom.num_patches_width],
                         orientations=[G.choose_orientation(pos)], lowres=lowres, patch_width=geom.patch_width,
                         group=group, device=device)[0]
    return G.stitch_canvas(out, pos, geom, background=zoomed_image.to(out[0].device), patch_size=out[0].shape[-1])
