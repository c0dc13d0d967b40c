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

The reference runs the stages 1 -> 2 -> 3 with a barrier (and a model reload) between them (:264-270).
Here a patch's stage s only waits for what it needs - its own stage s-1 output and the stage-s outputs
of its three neighbours - so the stages are PIPELINED: generalised wave g holds the stage-s tasks of
patch wave g - (s - 1) for every stage (`stage_waves`), 2n+1 steps for an n x n grid through three
stages instead of 3 (2n - 1), and the light stage-1/2 tasks fill ranks that the anti-diagonal leaves idle.
Inside a generalised wave the tasks are dealt to the ranks heaviest first with column affinity (a patch's
`above` neighbour was sampled on the same rank when the deal is balanced; `assign_tasks`).

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


STask = Tuple[int, int, int, int]  # (stage, canvas, i, j)
# relative cost of one patch per stage (batch-1 step time on MI355X x the reference's default timesteps
# 1024 / 256 / 256, profiles/README.md); only the ORDER of the deal depends on it
DEFAULT_STAGE_COST = {1: 38.8, 2: 7.5, 3: 68.0}


def stage_waves(per_canvas_pos: Sequence[Sequence[G.Pos]], orientations: Sequence[int], stages: Sequence[int],
                pipeline: bool = True) -> List[List[STask]]:
    """Generalised waves over (stage, canvas, patch).  A task needs its own previous-stage output and the same-stage
    outputs of its neighbours, so stage number k (0-based within `stages`) of patch wave w can run at step w + k.
    pipeline=False gives the reference's order: all waves of a stage, then the next stage."""
    waves = merged_waves(per_canvas_pos, orientations)
    out: List[List[STask]] = []
    if not pipeline:
        for st in stages:
            out.extend([[(st, c, i, j) for (c, i, j) in w] for w in waves])
        return out
    for g in range(len(waves) + len(stages) - 1):
        cur: List[STask] = []
        for k, st in enumerate(stages):
            if 0 <= g - k < len(waves):
                cur.extend((st, c, i, j) for (c, i, j) in waves[g - k])
        out.append(cur)
    return out


def assign_tasks(wave: Sequence[STask], world: int, stage_cost: Optional[Dict[int, float]] = None) -> List[List[STask]]:
    """Deal of one generalised wave: heaviest tasks first, each to the least-loaded rank, preferring rank
    (j + canvas) % world (column affinity: the patch above sat on the same rank).  Deterministic on every rank."""
    cost = dict(DEFAULT_STAGE_COST)
    cost.update(stage_cost or {})
    order = sorted(wave, key=lambda t: (-cost.get(t[0], 1.0), t[1], t[2], t[3]))
    load = [0.0] * world
    out: List[List[STask]] = [[] for _ in range(world)]
    for t in order:
        pref = (t[3] + t[1]) % world
        best = min(load)
        r = pref if load[pref] <= best + 1e-9 else load.index(best)
        out[r].append(t)
        load[r] += cost.get(t[0], 1.0)
    return out


def _all_gather_start(mine: torch.Tensor, counts: List[int], group):
    """All-gather of per-rank patch slabs with unequal counts (padded to the max count), started asynchronously:
    returns a function that waits and gives the per-rank slabs."""
    world = len(counts)
    if world == 1:
        return lambda: [mine]
    mx = max(counts)
    shape = (mx,) + tuple(mine.shape[1:])
    send = torch.zeros(shape, device=mine.device, dtype=mine.dtype)
    if mine.shape[0]:
        send[: mine.shape[0]] = mine
    recv = [torch.empty_like(send) for _ in range(world)]
    work = dist.all_gather(recv, send, group=group, async_op=True)

    def finish():
        work.wait()
        return [r[:c] for r, c in zip(recv, counts)]

    return finish


def _all_gather_patches(mine: torch.Tensor, counts: List[int], group) -> List[torch.Tensor]:
    return _all_gather_start(mine, counts, group)()


_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _run_stage_groups(sample_fn: Callable, groups, results: Dict[int, List[torch.Tensor]], device) -> None:
    """Runs the (stage, tasks, ...) groups of one generalised wave on this rank.  With a CUDA `device` and more than
    one group: the first (heaviest) on the calling thread and stream, the others from a second thread on a side
    stream (torch's current stream is per thread), both drained before the results are used."""
    # A (stage, batch) the sampler has not run yet would build its plan and capture its graph on the side thread
    # while this thread launches: such a wave runs its groups one after the other
    is_warm = getattr(sample_fn, "is_warm", None)
    cold = is_warm is not None and not all(is_warm(g[0], len(g[1])) for g in groups)
    if device is None or len(groups) < 2 or torch.device(device).type != "cuda" or cold:
        for g in groups:
            results[g[0]] = list(sample_fn(*g))
        return
    import threading

    # An unseeded sampler draws its Philox keys from torch's global CPU generator; two threads drawing from it would
    # make the patch -> seed mapping depend on their interleaving.  The draws happen HERE, on the calling thread, in
    # group order, and are handed in (torch.manual_seed then reproduces an overlapped run)
    seeded = getattr(sample_fn, "takes_base_seed", False)
    base = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if seeded else None for _ in groups]

    def call(n):
        return list(sample_fn(*groups[n], base_seed=base[n]) if seeded else sample_fn(*groups[n]))

    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))   # inputs assembled on the caller's stream
    errors = []

    def light():
        try:
            with torch.cuda.device(dev), torch.cuda.stream(side):
                for n in range(1, len(groups)):
                    results[groups[n][0]] = call(n)
            side.synchronize()
        except BaseException as e:   # re-raised on the calling thread
            errors.append(e)

    th = threading.Thread(target=light, name="kd-grid-light-stages")
    th.start()
    try:
        results[groups[0][0]] = call(0)
    finally:
        th.join()
    if errors:
        raise errors[0]
    torch.cuda.current_stream(dev).wait_stream(side)


def sample_grids(sample_fn: Callable, stages: Sequence[int], patch_pos: Sequence[Sequence[G.Pos]],
                 cond_images: Sequence[Optional[torch.Tensor]], overlap: float,
                 num_patches_width: Sequence[int], orientations: Optional[Sequence[int]] = None,
                 lowres: Optional[Sequence[Optional[torch.Tensor]]] = None,
                 patch_width: Optional[int] = None, group=None,
                 device: Optional[torch.device] = None, pipeline: bool = True,
                 stage_cost: Optional[Dict[int, float]] = None,
                 overlap_stages: Optional[torch.device] = None) -> List[List[torch.Tensor]]:
    """Runs `stages` (e.g. (1,2,3)) over one or more canvases and returns, on every rank,
    `out[c][idx]` = (3,S,S) final-stage patch `idx` of canvas c (index order of patch_pos[c]).

    cond_images[c]: (N_c, Cc, 1024, 1024) or None; lowres[c]: optional (N_c,3,s,s) start images for
    the first stage in `stages` (the reference's --ignore_unet_1 path, sample_ultra_res.py:417-420).
    pipeline=False keeps the reference's barrier between the stages (same results: every task sees the same
    inputs in either order).  overlap_stages=<the sampler's CUDA device> (pipelined waves): the stage groups of one generalised wave
    on this rank are independent, so the heaviest (stage 3: kernels that fill the chip) runs on the caller's stream
    while the lighter ones (batch-1 stage-1/2 passes: weight-bandwidth and launch-latency bound, most of the chip
    idle) run from a second host thread on a side stream; same results, the wave ends when both are done.  The
    sample_fn must have been warmed (plans built, graphs captured) and keep per-stage state apart."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ncanvas = len(patch_pos)
    stages = list(stages)
    orientations = list(orientations) if orientations is not None else [G.choose_orientation(p) for p in patch_pos]
    index = [{p: n for n, p in enumerate(pos)} for pos in patch_pos]
    start_low = [None] * ncanvas if lowres is None else [None if l is None else list(l) for l in lowres]
    # done[stage][canvas][(i, j)] = finished (3,S,S) patch, identical on every rank after the wave's exchange
    done: Dict[int, List[Dict[G.Pos, torch.Tensor]]] = {st: [dict() for _ in range(ncanvas)] for st in stages}
    prev_stage = {st: (stages[k - 1] if k > 0 else None) for k, st in enumerate(stages)}

    for wave in stage_waves(patch_pos, orientations, stages, pipeline):
        parts = assign_tasks(wave, world, stage_cost)
        mine = parts[rank]
        wave_stages = sorted({t[0] for t in wave})
        results: Dict[int, List[torch.Tensor]] = {}
        groups = []
        for st in sorted({t[0] for t in mine}, reverse=True):   # heaviest stage first
            S = G.PATCH_SIZES[st]
            tasks = [t for t in mine if t[0] == st]
            lows, conds, ips, ims = [], [], [], []
            for (_, c, i, j) in tasks:
                idx = index[c][(i, j)]
                cond = None if cond_images[c] is None else cond_images[c][idx]
                ip, im = G.assemble_inpaint((i, j), patch_pos[c], done[st][c], S, overlap, orientations[c],
                                            num_patches_width[c], cond_image=cond, patch_width=patch_width)
                ps = prev_stage[st]
                low = done[ps][c][(i, j)] if ps is not None else (None if start_low[c] is None else start_low[c][idx])
                lows.append(low)
                conds.append(cond)
                ips.append(ip)
                ims.append(im)
            groups.append((st, [(c, i, j) for (_, c, i, j) in tasks], lows, conds, ips, ims))
        _run_stage_groups(sample_fn, groups, results, overlap_stages)
        # one all-gather per stage present in the wave, all in flight together
        pending = []
        for st in wave_stages:
            S = G.PATCH_SIZES[st]
            outs = results.get(st, [])
            dev = device if device is not None else (outs[0].device if outs else torch.device("cpu"))
            slab = torch.stack(outs).to(dev) if outs else torch.zeros((0, 3, S, S), device=dev)
            per_rank = [[t for t in p if t[0] == st] for p in parts]
            pending.append((st, per_rank, _all_gather_start(slab.float().contiguous(), [len(p) for p in per_rank], group)))
        for st, per_rank, finish in pending:
            gathered = finish()
            for r, tasks in enumerate(per_rank):
                for n, (_, c, i, j) in enumerate(tasks):
                    done[st][c][(i, j)] = gathered[r][n]
    last = stages[-1]
    return [[done[last][c][p] for p in patch_pos[c]] for c in range(ncanvas)]


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

    warmed = set()   # (stage, batch) whose plan exists and whose step graph has been captured

    def fn(stage, tasks, lows, conds, ips, ims, base_seed=None):
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
            elif base_seed is not None:   # drawn by the scheduler on its own thread (overlapped stage groups)
                kw["seed"] = (base_seed + n0) % (2 ** 31 - 1)
            # the reference passes the (possibly all-zero) inpaint tensors for every grid patch (:149-174)
            kw.update(inpaint_images=stack(ips[sl]), inpaint_masks=stack(ims[sl]),
                      inpaint_resample_times=inpaint_resample)
            out = imagen.sample(**kw)
            warmed.add((stage, b))
            outs.extend(out[i] for i in range(b))
        return outs

    def is_warm(stage, ntasks):
        step = cap(stage)
        sizes = {min(step, ntasks - n0) for n0 in range(0, ntasks, step)}
        return all((stage, b) in warmed for b in sizes)

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
    fn.is_warm = is_warm
    fn.takes_base_seed = seed is None
    fn._force_batch = None
    return fn
