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

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

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


def generate_mag0_image(sample_fn: Callable, stages: Sequence[int] = (1, 2, 3), group=None,
                        device: Optional[torch.device] = None) -> torch.Tensor:
    """`generate_image(0, args)` (sample_ultra_res.py:264-270, :463): ONE unconditional sample through the stages -
    no conditioning image, no patch position and therefore no inpainting tensors (:88-91, :149).  Under
    torch.distributed rank 0 samples and broadcasts, so every rank starts the grid levels from the same image.
    `device`: where the image lives on every rank.  The broadcast needs the SAME kind of tensor on every rank: device
    memory under RCCL (backend "nccl" has no CPU tensors), host memory under gloo - so the ranks that did not sample
    allocate their receive buffer there, and rank 0 moves its sample there first."""
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    img = None
    if not multi or dist.get_rank(group) == 0:
        for st in stages:
            img = sample_fn(st, [(0, 0, 0)], [img], [None], [None], [None])[0]
    if multi:
        shape = (3, G.PATCH_SIZES[stages[-1]], G.PATCH_SIZES[stages[-1]])
        if dist.get_backend(group) == "gloo":
            xdev = torch.device("cpu")
        elif device is not None:
            xdev = torch.device(device)
        elif img is not None and img.is_cuda:
            xdev = img.device
        else:
            xdev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.empty(shape, dtype=torch.float32, device=xdev) if img is None else img.to(xdev).float().contiguous()
        dist.broadcast(buf, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        img = buf
    if device is not None:
        img = img.to(device)
    return img[None]


def generate_all_levels(sample_fns: Dict[int, Callable], overlap: float = 0.25, version: str = "ultra",
                        ignore_unet_1: bool = False, group=None, device: Optional[torch.device] = None,
                        patch_filter: Optional[Callable[[int, List[G.Pos]], List[G.Pos]]] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The reference's `main()` chain (sample_ultra_res.py:463-469): mag 0 (one unconditional 1024-px sample) ->
    mag 1 (grid over it; kidney: 8x8, canvas 6400^2) -> mag 2 (grid over the mag-1 canvas filtered by the tissue
    mask; kidney: 53x53 candidates, canvas 40960^2).  `sample_fns[level]` samples with that level's models (the
    reference loads `--unet{n}_mag{level}` checkpoints, :36-63).  `patch_filter(level, positions)` may thin a level's
    positions (tests, partial regeneration).  Returns the three images, each (1,3,W,W)."""
    mag0 = generate_mag0_image(sample_fns[0], group=group, device=device)
    out = [mag0]
    for level in (1, 2):
        pos = None
        if patch_filter is not None:
            pos = patch_filter(level, level_patches(out[-1], level, overlap, version)[1])
        out.append(generate_high_res_image(sample_fns[level], out[-1], level, overlap=overlap, version=version,
                                           ignore_unet_1=ignore_unet_1, group=group, device=device, patch_pos=pos))
    return tuple(out)
