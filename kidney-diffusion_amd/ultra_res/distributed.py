"""Multi-GPU scheduling of the ultra-res patch grid over `torch.distributed` ranks.

The reference spawns one process per GPU that pull patches from a shared `mp.Queue`, spin until a
patch's three neighbours are done and publish results through a `Manager().dict()` in host memory
(sample_ultra_res.py:213-261).  Here the dependency structure is made explicit instead:

  * patches are grouped into anti-diagonal WAVES (grid.wavefronts); inside a wave they are
    independent and are dealt to the ranks (one process per GPU);
  * what a patch needs from a finished neighbour is three overlap STRIPS (sample_ultra_res.py:156-170:
    the bottom rows of the patch above, the facing columns of the patch beside it, the facing corner of
    the diagonal one; <= 3.1 MB each at stage 3) - so a finished patch stays on the rank that sampled it and
    only those strips travel, point to point, to the <= 3 ranks that consume them (`ExchangePlan`: the whole
    deal is a pure function of the grid, so every rank knows every message in advance).  When its wave is done a
    rank posts the sends and receives of that wave, one asynchronous batch per peer in an order all ranks share,
    and waits for a bundle only when a task that needs it is about to start: there is no barrier between waves
    (RCCL send/recv over xGMI on the GPU node, gloo in the CPU tests).  Under RCCL a batch is one group kernel
    that completes when the peer has posted its side and the batches of one rank run in posting order, so a
    consumer waits for its producer and for the peers of the batches this rank posted before that one - not for
    every peer of the wave; `ExchangePlan.makespan` does not model that residual coupling (free exchange);
  * ONE all-gather of equal-sized slabs after the last wave gives every rank the final-stage patches of the
    whole canvas (the north star's "RCCL all-gather to reassemble the stitched canvas");
  * several canvases can be scheduled together: their waves are merged, which is what lifts the
    8x8-grid bound of 64/15 = 4.27x on 8 GPUs.

The reference runs the stages 1 -> 2 -> 3 with a barrier (and a model reload) between them (:264-270).
Here a patch's stage s only waits for what it needs - its own stage s-1 output and the stage-s strips
of its three neighbours - so the stages are PIPELINED: generalised wave g holds the stage-s tasks of
patch wave g - (s - 1) for every stage (`stage_waves`), 2n+1 steps for an n x n grid through three
stages instead of 3 (2n - 1), and the light stage-1/2 tasks fill ranks that the anti-diagonal leaves idle.
Inside a generalised wave the tasks are dealt to the ranks heaviest first with column affinity (a patch's
`above` neighbour was sampled on the same rank when the deal is balanced; `assign_tasks`).

`sample_fn(stage, idxs, lowres, cond, inpaint_patch, inpaint_mask) -> (n,3,S,S)` does the actual
sampling (the HIP engine through `Imagen.sample` in production, a deterministic stub in the tests).
"""
from __future__ import annotations

import time
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


Item = Tuple[str, STask]   # (kind, producer): kind in grid.STRIP_KINDS or "low" (the whole previous-stage patch)


class ExchangePlan:
    """Everything about a run that does not depend on the data: the generalised waves, who samples what, which
    strips each task needs from whom, and the point-to-point bundles of every wave.  Built identically on every rank.

      waves[g]            tasks of generalised wave g
      parts[g][r]         the tasks of wave g dealt to rank r (in the order they are handed to sample_fn)
      owner[t], wave_of[t]
      needs[t]            [(kind, producer task)] for the neighbours / previous stage that exist
      bundles[g][(src, dst)]   [(kind, producer)] produced on src in wave g and consumed on dst, in send order
    """

    def __init__(self, patch_pos, orientations, stages, world, pipeline=True, stage_cost=None):
        self.world = world
        self.stages = list(stages)
        self.waves = stage_waves(patch_pos, orientations, self.stages, pipeline)
        self.parts = [assign_tasks(w, world, stage_cost) for w in self.waves]
        self.owner: Dict[STask, int] = {}
        self.wave_of: Dict[STask, int] = {}
        for g, parts in enumerate(self.parts):
            for r, p in enumerate(parts):
                for t in p:
                    self.owner[t], self.wave_of[t] = r, g
        prev = {st: (self.stages[k - 1] if k > 0 else None) for k, st in enumerate(self.stages)}
        sets = [set(p) for p in patch_pos]
        self.needs: Dict[STask, List[Item]] = {}
        self.bundles: List[Dict[Tuple[int, int], List[Item]]] = [dict() for _ in self.waves]
        for g, parts in enumerate(self.parts):
            for r, p in enumerate(parts):
                for t in p:
                    st, c, i, j = t
                    need: List[Item] = []
                    for kind, q in G.neighbour_positions((i, j), orientations[c]).items():
                        if q in sets[c]:
                            need.append((kind, (st, c) + q))
                    if prev[st] is not None:
                        need.append(("low", (prev[st], c, i, j)))
                    self.needs[t] = need
        # bundles in a canonical order: by producing wave, then the consumer's place in its own wave
        for g, parts in enumerate(self.parts):
            for r, p in enumerate(parts):
                for t in p:
                    for item in self.needs[t]:
                        src = self.owner[item[1]]
                        assert self.wave_of[item[1]] < g, "a task depends on a task of its own or a later wave"
                        if src != r:
                            self.bundles[self.wave_of[item[1]]].setdefault((src, r), []).append(item)

    def item_numel(self, item: Item, overlap: float) -> int:
        kind, (st, _, _, _) = item
        S = G.PATCH_SIZES[st]
        if kind == "low":
            return 3 * S * S
        sh = G.strip_shape(kind, S, int(overlap * S))
        return sh[0] * sh[1] * sh[2]

    def p2p_bytes(self, overlap: float) -> int:
        """fp32 bytes that cross between ranks before the final gather (all bundles of all waves)."""
        return 4 * sum(self.item_numel(it, overlap) for b in self.bundles for items in b.values() for it in items)

    def p2p_messages(self) -> int:
        return sum(len(b) for b in self.bundles)

    def makespan(self, stage_cost: Dict[int, float]) -> float:
        """Length of the schedule when nothing but the dependencies holds a rank back (no barrier between waves, free
        exchange): a rank starts its share of wave g when it has finished wave g - 1 and every producer of that
        share is done; the tasks of the share run one after the other."""
        finish: Dict[STask, float] = {}
        free = [0.0] * self.world
        for g, parts in enumerate(self.parts):
            for r, p in enumerate(parts):
                if not p:
                    continue
                t0 = max([free[r]] + [finish[it[1]] for t in p for it in self.needs[t]])
                for t in p:
                    t0 += stage_cost.get(t[0], 1.0)
                    finish[t] = t0
                free[r] = t0
        return max(free)


_connected = set()


def _connect_all(group, device, world: int, rank: int) -> None:
    """RCCL sets up a send / recv connection the first time a pair uses it, inside the group call that carries the
    operation; in the run itself the ranks reach their exchange calls at different times.  One synchronised batch in
    which every rank exchanges a word with every other rank (all ranks enter it together, right after the caller's
    barrier or at the first grid) leaves only enqueueing to the later calls.  Done once per process and group."""
    key = (id(group), world)
    if world < 2 or key in _connected or dist.get_backend(group) != "nccl":
        return
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    tx = torch.zeros(world, device=device)
    rx = torch.empty(world, device=device)
    ops = []
    for r in range(world):
        if r != rank:
            ops.append(dist.P2POp(dist.isend, tx[r:r + 1], glob(r), group))
            ops.append(dist.P2POp(dist.irecv, rx[r:r + 1], glob(r), group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize(device)
    _connected.add(key)


def connect_ranks(group=None, device: Optional[torch.device] = None) -> None:
    """Sets up the point-to-point connections of `sample_grids` ahead of the first grid (call it on every rank, e.g. behind
    the warm-up): otherwise the first exchange of a process pays for them."""
    if dist.is_initialized() and device is not None and torch.device(device).type == "cuda":
        _connect_all(group, torch.device(device), dist.get_world_size(group), dist.get_rank(group))


def _cut_item(item: Item, patch: torch.Tensor, overlap: float, orientation: int) -> torch.Tensor:
    kind = item[0]
    if kind == "low":
        return patch
    return G.cut_strip(kind, patch, int(overlap * patch.shape[-1]), orientation)


def _item_shape(item: Item, overlap: float) -> Tuple[int, int, int]:
    kind, (st, _, _, _) = item
    S = G.PATCH_SIZES[st]
    return (3, S, S) if kind == "low" else G.strip_shape(kind, S, int(overlap * S))


_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _run_stage_groups(sample_fn: Callable, groups, results: Dict[int, List[torch.Tensor]], device) -> None:
    """Runs the (stage, tasks, ...) groups of one generalised wave on this rank.  With a CUDA `device` and more than
    one group: the first (heaviest) on the calling thread and stream, the others from a second thread on a side
    stream (torch's current stream is per thread), both drained before the results are used."""
    # A (stage, batch) the sampler has not run yet would build its plan and capture its graph on the side thread
    # while this thread launches: such a wave runs its groups one after the other
    # An unseeded sampler draws its Philox keys from torch's global CPU generator; two threads drawing from it would
    # make the patch -> seed mapping depend on their interleaving, and a sequential wave must not consume the generator
    # differently from an overlapped one.  So the draws ALWAYS happen here, on the calling thread, one per group in
    # group order, and are handed in: torch.manual_seed reproduces a run whether or not its waves overlapped
    seeded = getattr(sample_fn, "takes_base_seed", False)
    base = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if seeded else None for _ in groups]

    def call(n):
        return list(sample_fn(*groups[n], base_seed=base[n]) if seeded else sample_fn(*groups[n]))

    is_warm = getattr(sample_fn, "is_warm", None)
    cold = is_warm is not None and not all(is_warm(g[0], len(g[1])) for g in groups)
    if device is None or len(groups) < 2 or torch.device(device).type != "cuda" or cold:
        for n, g in enumerate(groups):
            results[g[0]] = call(n)
        return
    import threading

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
                 overlap_stages: Optional[torch.device] = None,
                 gather: str = "all", stats: Optional[dict] = None) -> List[List[Optional[torch.Tensor]]]:
    """Runs `stages` (e.g. (1,2,3)) over one or more canvases and returns `out[c][idx]` = (3,S,S) final-stage
    patch `idx` of canvas c (index order of patch_pos[c]) - on every rank with gather="all" (one all-gather of
    equal-sized slabs after the last wave), on rank 0 alone with gather="root" (one gather; the other ranks keep
    their own patches, None elsewhere), only this rank's own patches with gather="none".

    cond_images[c]: (N_c, Cc, 1024, 1024) or None; lowres[c]: optional (N_c,3,s,s) start images for
    the first stage in `stages` (the reference's --ignore_unet_1 path, sample_ultra_res.py:417-420).
    pipeline=False keeps the reference's barrier between the stages (same results: every task sees the same
    inputs in either order).  overlap_stages=<the sampler's CUDA device> (pipelined waves): the stage groups of one generalised wave
    on this rank are independent, so the heaviest (stage 3: kernels that fill the chip) runs on the caller's stream
    while the lighter ones (batch-1 stage-1/2 passes: weight-bandwidth and launch-latency bound, most of the chip
    idle) run from a second host thread on a side stream; same results, the wave ends when both are done.  The
    sample_fn must have been warmed (plans built, graphs captured) and keep per-stage state apart.
    `device`: where finished patches and strips are kept and exchanged (HBM under RCCL, host memory under gloo).
    `stats` (a dict) receives what the exchange moved: p2p bytes / messages of the whole job and of this rank,
    the bytes of the final gather and the number of blocking collectives."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ncanvas = len(patch_pos)
    stages = list(stages)
    orientations = list(orientations) if orientations is not None else [G.choose_orientation(p) for p in patch_pos]
    index = [{p: n for n, p in enumerate(pos)} for pos in patch_pos]
    start_low = [None] * ncanvas if lowres is None else [None if l is None else list(l) for l in lowres]
    plan = ExchangePlan(patch_pos, orientations, stages, world, pipeline, stage_cost)
    if world > 1 and device is not None and torch.device(device).type == "cuda":
        _connect_all(group, torch.device(device), world, rank)
    local: Dict[STask, torch.Tensor] = {}     # finished patches sampled on this rank
    have: Dict[Item, torch.Tensor] = {}       # strips / previous-stage patches received from other ranks
    pending: Dict[Tuple[int, int], tuple] = {}   # (src, wave) -> (works, buffer, items) of a bundle in flight
    send_keep = []                             # (works, buffer) of posted sends
    sent_bytes = 0

    waited = {}   # id -> handle (kept alive: ids are reused after collection): a work handle is waited for ONCE - a
                  # coalesced RCCL batch shares one handle between its bundles, and a second wait on a gloo handle hangs

    def wait_all(works):
        for w in works:
            if id(w) not in waited:
                waited[id(w)] = w
                w.wait()

    # how long this rank stood in receive waits: host time (gloo: the wait blocks the thread) and, where the exchange
    # lives in HBM, the stall of the stream between two events around the wait (RCCL: the wait is a stream dependency)
    x_cuda = world > 1 and device is not None and torch.device(device).type == "cuda"
    wait_host_s = 0.0
    wait_events = []
    batches_posted = 0
    idle_waves = 0

    def timed_wait(works):
        nonlocal wait_host_s
        ev0 = None
        if x_cuda:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(torch.device(device)))
        t0 = time.perf_counter()
        wait_all(works)
        wait_host_s += time.perf_counter() - t0
        if x_cuda:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream(torch.device(device)))
            wait_events.append((ev0, ev1))

    def receive(src: int, g: int):
        rec = pending.pop((src, g), None)
        if rec is None:
            return
        works, buf, items = rec
        timed_wait(works)
        off = 0
        for it in items:
            sh = _item_shape(it, overlap)
            n = sh[0] * sh[1] * sh[2]
            have[it] = buf[off:off + n].view(sh)
            off += n

    def fetch(item: Item) -> torch.Tensor:
        prod = item[1]
        if plan.owner[prod] == rank:
            return _cut_item(item, local[prod], overlap, orientations[prod[1]])
        receive(plan.owner[prod], plan.wave_of[prod])
        return have.pop(item)

    for g, wave in enumerate(plan.waves):
        mine = plan.parts[g][rank]
        if not mine and wave:
            idle_waves += 1   # a generalised wave in which other ranks sample and this one has nothing to do
        results: Dict[int, List[torch.Tensor]] = {}
        groups = []
        for st in sorted({t[0] for t in mine}, reverse=True):   # heaviest stage first
            S = G.PATCH_SIZES[st]
            ov = int(overlap * S)
            tasks = [t for t in mine if t[0] == st]
            lows, conds, ips, ims = [], [], [], []
            for t in tasks:
                _, c, i, j = t
                idx = index[c][(i, j)]
                cond = None if cond_images[c] is None else cond_images[c][idx]
                strips = G.fallback_strips((i, j), patch_pos[c], S, overlap, orientations[c], num_patches_width[c],
                                           cond, patch_width)
                low = None if start_low[c] is None else start_low[c][idx]
                for item in plan.needs[t]:
                    if item[0] == "low":
                        low = fetch(item)
                    else:
                        strips[item[0]] = fetch(item)
                ip, im = G.inpaint_from_strips(strips, S, ov, orientations[c])
                lows.append(low)
                conds.append(cond)
                ips.append(ip)
                ims.append(im)
            groups.append((st, [(c, i, j) for (_, c, i, j) in tasks], lows, conds, ips, ims))
        _run_stage_groups(sample_fn, groups, results, overlap_stages)
        for grp in groups:
            st = grp[0]
            for (c, i, j), o in zip(grp[1], results[st]):
                dev = device if device is not None else o.device
                local[(st, c, i, j)] = o.to(dev).float()
        if world == 1:
            continue
        # this wave's exchange, not waited for here: ONE asynchronous batch per peer (my strips to it and the strips it
        # produced in this wave that tasks of mine will consume later).  Under RCCL a batch is one group kernel that
        # completes when the peer has posted its side, so a batch per PEER lets a consumer wait for its producer
        # alone - a single batch per wave would complete with its slowest peer.  Every rank posts its batches in the
        # same total order (wave, low rank, high rank): the globally smallest unfinished batch is at the head of both
        # its ranks' queues, so the order cannot deadlock (groups on one communicator run in posting order)
        by_pair: Dict[Tuple[int, int], list] = {}
        for (src, dst), items in plan.bundles[g].items():
            if rank in (src, dst):
                by_pair.setdefault((min(src, dst), max(src, dst)), []).append((src, dst, items))
        for pair in sorted(by_pair):
            ops, recvs, sends = [], [], []
            for src, dst, items in sorted(by_pair[pair], key=lambda e: e[:2]):   # low -> high first, on both sides
                if src == rank:
                    flat = torch.cat([_cut_item(it, local[it[1]], overlap, orientations[it[1][1]]).reshape(-1)
                                      for it in items])
                    sent_bytes += flat.numel() * 4
                    ops.append(dist.P2POp(dist.isend, flat, dst if group is None else dist.get_global_rank(group, dst), group))
                    sends.append(flat)
                else:
                    n = sum(plan.item_numel(it, overlap) for it in items)
                    ref = next(iter(local.values())) if local else None
                    dev = device if device is not None else (ref.device if ref is not None else torch.device("cpu"))
                    buf = torch.empty(n, device=dev, dtype=torch.float32)
                    ops.append(dist.P2POp(dist.irecv, buf, src if group is None else dist.get_global_rank(group, src), group))
                    recvs.append((src, buf, items))
            works = dist.batch_isend_irecv(ops)
            batches_posted += 1
            if len(works) == len(ops):   # one handle per operation (gloo)
                k = 0
                for op, w in zip(ops, works):
                    if op.op is dist.irecv:
                        src, buf, items = recvs[k]
                        k += 1
                        pending[(src, g)] = ([w], buf, items)
                    else:
                        send_keep.append(([w], op.tensor))
            else:                        # one handle for the coalesced batch (RCCL): stream-ordered, not host-blocking
                for src, buf, items in recvs:
                    pending[(src, g)] = (works, buf, items)
                send_keep.append((works, sends))
    for works, _ in send_keep:
        wait_all(works)
    assert not pending, "a posted bundle was never consumed"
    last = stages[-1]
    S = G.PATCH_SIZES[last]
    own = [[local.get((last, c) + p) for p in patch_pos[c]] for c in range(ncanvas)]
    gather_bytes = 0
    if world > 1 and gather in ("all", "root"):
        # ONE collective per job: equal-sized slabs of final-stage patches, canvas-major in index order per rank
        per_rank = [[(c, p) for c in range(ncanvas) for p in patch_pos[c] if plan.owner[(last, c) + p] == r]
                    for r in range(world)]
        mx = max(len(p) for p in per_rank)
        ref = next((t for row in own for t in row if t is not None), None)
        dev = device if device is not None else (ref.device if ref is not None else torch.device("cpu"))
        slab = torch.zeros((mx, 3, S, S), device=dev, dtype=torch.float32)
        for n, (c, p) in enumerate(per_rank[rank]):
            slab[n] = local[(last, c) + p]
        if gather == "all":
            allp = torch.empty((world * mx, 3, S, S), device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(allp, slab, group=group)
            parts_ = [allp[r * mx:(r + 1) * mx] for r in range(world)]
        else:   # only rank 0 of the group stitches: a gather moves 1 / world of the all-gather's bytes
            parts_ = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
            dist.gather(slab, parts_, dst=0 if group is None else dist.get_global_rank(group, 0), group=group)
        if parts_ is not None:
            gather_bytes = world * slab.numel() * 4
            for r, lst in enumerate(per_rank):
                for n, (c, p) in enumerate(lst):
                    own[c][index[c][p]] = parts_[r][n]
    if stats is not None:
        wait_stream_s = 0.0
        if wait_events:
            torch.cuda.synchronize(torch.device(device))
            wait_stream_s = sum(a.elapsed_time(b) for a, b in wait_events) * 1e-3
        stats.update(recv_wait_host_s_this_rank=wait_host_s, recv_wait_stream_s_this_rank=wait_stream_s,
                     recv_waits_this_rank=len(wait_events) if x_cuda else None, idle_waves_this_rank=idle_waves,
                     tasks_this_rank=sum(len(p[rank]) for p in plan.parts), p2p_batches_posted_by_this_rank=batches_posted)
        stats.update(p2p_bytes_total=plan.p2p_bytes(overlap), p2p_messages_total=plan.p2p_messages(),
                     p2p_bytes_sent_by_this_rank=sent_bytes, final_gather_bytes_per_rank=gather_bytes,
                     blocking_collectives=int(world > 1 and gather in ("all", "root")), waves=len(plan.waves),
                     whole_patch_allgather_bytes_per_rank=sum(
                         4 * 3 * G.PATCH_SIZES[t[0]] ** 2 * (world - 1) for w in plan.waves for t in w) if world > 1 else 0)
    return own


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
            # the reference passes the (possibly all-zero) inpaint tensors for every grid patch (:149-174) and none
            # for the single mag-0 image, which has no position (:88-91)
            if ips[n0] is not None:
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
