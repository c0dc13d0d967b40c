"""Multi-GPU scheduling of the ultra-res patch grid over `torch.distributed` ranks.

The reference spawns one process per GPU that pull patches from a shared `mp.Queue`, spin until a
patch's three neighbours are done and publish results through a `Manager().dict()` in host memory
(sample_ultra_res.py:213-261).  Here the dependency structure is made explicit instead:

  * patches are grouped into anti-diagonal WAVES (grid.wavefronts); inside a wave they are
    independent and are dealt round-robin to the ranks (one process per GPU);
  * after a wave, every rank contributes its finished patches to ONE all-gather (RCCL over xGMI on
    the GPU node, gloo in the CPU tests) so that each rank holds the neighbours it needs for the
    next wave — and, after the last wave, the whole canvas;
  * several canvases can be scheduled together: their waves are merged, which is what lifts the
    8x8-grid bound of 64/15 = 4.27x on 8 GPUs.

Stages run 1 -> 2 -> 3 with a barrier between them, as the reference does (:264-270).
`sample_fn(stage, idxs, lowres, cond, inpaint_patch, inpaint_mask) -> (n,3,S,S)` does the actual
sampling (the HIP engine through `Imagen.sample` in production, a deterministic stub in the tests).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import grid as G

Task = Tuple[int, int, int]  # (canvas, i, j)


def merged_waves(per_canvas_pos: Sequence[Sequence[G.Pos]], orientations: Sequence[int]) -> List[List[Task]]:
    waves: List[List[Task]] = []
    for c, (pos, o) in enumerate(zip(per_canvas_pos, orientations)):
        for k, w in enumerate(G.wavefronts(pos, o)):
            while len(waves) <= k:
                waves.append([])
            waves[k].extend((c, i, j) for i, j in w)
    return waves


def assign(wave: Sequence[Task], world: int) -> List[List[Task]]:
    """Round-robin deal of one wave's tasks to ranks; deterministic on every rank."""
    out: List[List[Task]] = [[] for _ in range(world)]
    for n, t in enumerate(wave):
        out[n % world].append(t)
    return out


def schedule_length(waves: Sequence[Sequence[Task]], world: int) -> int:
    """Number of sequential patch slots (equal-cost patches): sum over waves of ceil(|wave| / world)."""
    return sum(-(-len(w) // world) for w in waves)


def _all_gather_patches(mine: torch.Tensor, counts: List[int], group) -> List[torch.Tensor]:
    """All-gather of per-rank patch slabs with unequal counts (padded to the max count)."""
    world = len(counts)
    if world == 1:
        return [mine]
    mx = max(counts)
    shape = (mx,) + tuple(mine.shape[1:])
    send = torch.zeros(shape, device=mine.device, dtype=mine.dtype)
    if mine.shape[0]:
        send[: mine.shape[0]] = mine
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    return [r[:c] for r, c in zip(recv, counts)]


def sample_grids(sample_fn: Callable, stages: Sequence[int], patch_pos: Sequence[Sequence[G.Pos]],
                 cond_images: Sequence[Optional[torch.Tensor]], overlap: float,
                 num_patches_width: Sequence[int], orientations: Optional[Sequence[int]] = None,
                 lowres: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 patch_width: Optional[int] = None, group=None,
                 device: Optional[torch.device] = None) -> List[List[torch.Tensor]]:
    """Runs `stages` (e.g. (1,2,3)) over one or more canvases and returns, on every rank,
    `out[c][idx]` = (3,S,S) final-stage patch `idx` of canvas c (index order of patch_pos[c]).

    cond_images[c]: (N_c, Cc, 1024, 1024) or None; lowres[c]: optional (N_c,3,s,s) start images for
    the first stage in `stages` (the reference's --ignore_unet_1 path, sample_ultra_res.py:417-420)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ncanvas = len(patch_pos)
    orientations = list(orientations) if orientations is not None else [G.choose_orientation(p) for p in patch_pos]
    index = [{p: n for n, p in enumerate(pos)} for pos in patch_pos]
    waves = merged_waves(patch_pos, orientations)
    prev: List[Optional[List[torch.Tensor]]] = [None] * ncanvas
    if lowres is not None:
        prev = [None if l is None else [t for t in l] for l in lowres]

    for stage in stages:
        S = G.PATCH_SIZES[stage]
        done: List[Dict[G.Pos, torch.Tensor]] = [dict() for _ in range(ncanvas)]
        for wave in waves:
            parts = assign(wave, world)
            mine = parts[rank]
            outs = []
            if mine:
                lows, conds, ips, ims = [], [], [], []
                for (c, i, j) in mine:
                    idx = index[c][(i, j)]
                    cond = None if cond_images[c] is None else cond_images[c][idx]
                    ip, im = G.assemble_inpaint((i, j), patch_pos[c], done[c], S, overlap, orientations[c],
                                                num_patches_width[c], cond_image=cond, patch_width=patch_width)
                    lows.append(None if prev[c] is None else prev[c][idx])
                    conds.append(cond)
                    ips.append(ip)
                    ims.append(im)
                res = sample_fn(stage, mine, lows, conds, ips, ims)
                outs = [r for r in res]
            dev = device if device is not None else (outs[0].device if outs else torch.device("cpu"))
            slab = torch.stack(outs).to(dev) if outs else torch.zeros((0, 3, S, S), device=dev)
            gathered = _all_gather_patches(slab.float().contiguous(), [len(p) for p in parts], group)
            for r, tasks in enumerate(parts):
                for n, (c, i, j) in enumerate(tasks):
                    done[c][(i, j)] = gathered[r][n]
        prev = [[done[c][p] for p in patch_pos[c]] for c in range(ncanvas)]
        if dist.is_initialized() and world > 1:
            dist.barrier(group=group)
    return prev  # type: ignore[return-value]


def outpaint_canvas(sample_fn: Callable, num_patches_width: int, overlap: float = 0.25, canvases: int = 1,
                    group=None, device: Optional[torch.device] = None) -> List[torch.Tensor]:
    """The unconditional outpainting driver of the reference (outpainting.py:173-243): an n x n grid of
    patches through stages 1 -> 2 -> 3, orientation -1, no conditioning images, each patch inpainted
    from its finished above / left / above-left neighbours, pasted onto a zero canvas of width
    1024 + (n-1)*int(1024*(1-overlap)).  Returns one (1,3,W,W) canvas per requested canvas."""
    n = num_patches_width
    pos = [(i, j) for i in range(n) for j in range(n)]
    out = sample_grids(sample_fn, (1, 2, 3), [pos] * canvases, [None] * canvases, overlap, [n] * canvases,
                       orientations=[-1] * canvases, group=group, device=device)
    P = out[0][0].shape[-1]  # 1024 with the reference's models
    stride = int(P * (1 - overlap))
    geom = G.GridGeometry(0, 0, n, stride, P + (n - 1) * stride)
    return [G.stitch_canvas(o, pos, geom, background=None, patch_size=P) for o in out]


def imagen_sample_fn(load_imagen: Callable, inpaint_resample: int, device: torch.device, use_graph: bool = True,
                     seed: Optional[int] = None, max_batch=1):
    """sample_fn over the drop-in `imagen_pytorch.Imagen` (HIP engine).  `load_imagen(stage)` returns
    the Imagen holding the real unet of that stage (the reference re-loads one stage at a time,
    sample_ultra_res.py:79; a cache keeps all three resident here).  With `max_batch` = 1 each patch
    is one `imagen.sample(batch_size=1, ...)` call with exactly the reference's kwargs (:183-195).

    `max_batch` (int, or {stage: int}) lets the patches of one wave that landed on this rank share a
    `sample()` call: they are independent (a wave holds no two neighbours), every patch keeps its own
    conditioning / low-res / inpaint tensors, and a batch-1 pass of the 64-px and 256-px UNets is
    weight-bandwidth bound, so a batch of 8 costs about as much as two single patches.  The plans of
    the different batch sizes share one packed-weight store (kd_unet_create_shared)."""
    cache = {}

    def cap(stage):
        m = max_batch.get(stage, 1) if isinstance(max_batch, dict) else max_batch
        return max(1, int(m))

    def stack(ts):
        return None if ts[0] is None else torch.stack([t.to(device) for t in ts])

    def fn(stage, tasks, lows, conds, ips, ims):
        if stage not in cache:
            cache[stage] = load_imagen(stage).to(device)
        imagen = cache[stage]
        outs = []
        step = fn._force_batch or cap(stage)
        for n0 in range(0, len(tasks), step):
            sl = slice(n0, n0 + step)
            b = len(tasks[sl])
            kw = dict(batch_size=b, return_pil_images=False, cond_images=stack(conds[sl]),
                      start_image_or_video=stack(lows[sl]), start_at_unet_number=stage, stop_at_unet_number=stage,
                      use_tqdm=False, device=device, use_graph=use_graph)
            if seed is not None:
                kw["seed"] = seed + 7919 * stage + 104729 * hash(tasks[n0]) % (2 ** 31)
            # the reference passes the (possibly all-zero) inpaint tensors for every grid patch (:149-174)
            kw.update(inpaint_images=stack(ips[sl]), inpaint_masks=stack(ims[sl]),
                      inpaint_resample_times=inpaint_resample)
            out = imagen.sample(**kw)
            outs.extend(out[i] for i in range(b))
        return outs

    def warm(batches_per_stage, cond_image=None):
        """Builds what a timed run must not pay for: for every stage its Imagen on the device and, for every batch
        size in `batches_per_stage[stage]`, the execution plan and the captured step graph (one sample() call on
        zero inpaint tensors).  Call it on EVERY rank before the first timed grid."""
        for stage, batches in batches_per_stage.items():
            S, s_prev = G.PATCH_SIZES[stage], G.PATCH_SIZES.get(stage - 1)
            for b in batches:
                tasks = [(0, 0, n) for n in range(b)]
                lows = [None if s_prev is None else torch.zeros(3, s_prev, s_prev, device=device) for _ in tasks]
                conds = [None if cond_image is None else cond_image for _ in tasks]
                ips = [torch.zeros(3, S, S, device=device) for _ in tasks]
                ims = [torch.zeros(S, S, device=device) for _ in tasks]
                fn._force_batch = b
                try:
                    fn(stage, tasks, lows, conds, ips, ims)
                finally:
                    fn._force_batch = None

    fn.warm = warm
    fn._force_batch = None
    return fn
