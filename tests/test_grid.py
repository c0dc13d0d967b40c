"""Patch-grid geometry / wavefront / stitch (CPU) and the world_size-2 gloo run of the distributed
scheduler against the single-process result."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from ultra_res import distributed as D
from ultra_res import grid as G


def test_geometry_matches_survey_numbers():
    # kidney mag1 -> patch_width 166, dist 124, 8x8, canvas 6400; mag2 -> 161, 120, 53x53
    g1 = G.grid_geometry(1024, 1, 0.25)
    assert (g1.patch_width, g1.patch_dist, g1.num_patches_width, g1.out_patch_dist, g1.canvas_width) == \
        (166, 124, 8, 768, 6400)
    g2 = G.grid_geometry(6400, 2, 0.25)
    assert (g2.patch_width, g2.patch_dist, g2.num_patches_width) == (161, 120, 53)
    ga = G.grid_geometry(1024, 1, 0.25, sizes=G.MAG_LEVEL_SIZES_AIRS, prefer_in_bounds=True)
    assert (ga.patch_width, ga.patch_dist, ga.num_patches_width, ga.canvas_width) == (340, 255, 3, 2560)


@pytest.mark.parametrize("o", [-1, 1])
def test_wavefront_order_and_dependency_bound(o):
    pos = G.grid_geometry(1024, 1, 0.25).positions
    waves = G.wavefronts(pos, o)
    assert len(waves) == 15 and sorted(len(w) for w in waves) == sorted([1, 2, 3, 4, 5, 6, 7, 8, 7, 6, 5, 4, 3, 2, 1])
    seen = set()
    for w in waves:
        for p in w:
            assert all(d in seen for d in G.dependencies(p, pos, o))
        seen |= set(w)
    first = (0, 0) if o == -1 else (0, 7)
    assert waves[0] == [first]
    tasks = D.merged_waves([pos], [o])
    # 64 / 36 / 22 / 15 sequential slots on 1 / 2 / 4 / 8 GPUs
    assert [D.schedule_length(tasks, g) for g in (1, 2, 4, 8)] == [64, 36, 22, 15]
    two = D.merged_waves([pos, pos], [o, o])
    assert D.schedule_length(two, 8) == 22   # two canvases in flight: 128/22 = 5.8x instead of 4.27x


def test_orientation_choice_and_filtered_grid():
    full = [(i, j) for i in range(4) for j in range(4)]
    assert G.choose_orientation(full) == 1  # tie -> +1, as the reference's strict '>' does
    tri = [(i, j) for i in range(4) for j in range(4) if j >= i]   # more free top-left corners
    assert G.choose_orientation([p for p in tri if p != (0, 0)]) in (-1, 1)
    waves = G.wavefronts(tri, 1)
    assert sum(len(w) for w in waves) == len(tri)


def test_inpaint_assembly_matches_reference_slicing():
    S, ov = 16, 4
    mk = lambda v: torch.full((3, S, S), float(v)) + torch.arange(S).float()[None, None, :] * 0.01 \
        + torch.arange(S).float()[None, :, None] * 0.1
    pos = [(i, j) for i in range(2) for j in range(2)]
    done = {(0, 0): mk(1), (0, 1): mk(2), (1, 0): mk(3)}
    patch, mask = G.assemble_inpaint((1, 1), pos, done, S, 0.25, -1, 2)
    assert torch.equal(patch[:, :ov, :], torch.cat((done[(0, 0)][:, -ov:, -ov:], done[(0, 1)][:, -ov:, ov:]), 2))
    assert torch.equal(patch[:, ov:, :ov], done[(1, 0)][:, ov:, -ov:])
    assert mask[:ov].all() and mask[:, :ov].all() and mask[ov:, ov:].sum() == 0
    assert patch[:, ov:, ov:].abs().sum() == 0
    # orientation +1 mirrors left/right
    done = {(0, 1): mk(1), (0, 0): mk(2), (1, 1): mk(3)}
    patch, mask = G.assemble_inpaint((1, 0), pos, done, S, 0.25, 1, 2)
    assert torch.equal(patch[:, ov:, -ov:], done[(1, 1)][:, ov:, :ov]) and mask[:, -ov:].all()
    assert torch.equal(patch[:, :ov, -ov:], done[(0, 1)][:, -ov:, :ov])
    # first patch: nothing known
    patch, mask = G.assemble_inpaint((0, 0), pos, {}, S, 0.25, -1, 2)
    assert patch.abs().sum() == 0 and mask.sum() == 0


def test_cond_images_and_stitch_semantics(monkeypatch):
    geom = G.GridGeometry(patch_width=8, patch_dist=6, num_patches_width=2, out_patch_dist=768, canvas_width=1792)
    z = torch.rand(1, 3, 14, 14)
    pos = geom.positions
    # an image narrower than PATCH_SIZE is zero-padded to it, as torchvision's CenterCrop does (:391)
    padded = G.cond_images_for_grid(z, geom, pos[:1], fill_color=0.95)
    assert padded.shape == (1, 3, 1024, 1024) and padded[0, :, :505].abs().sum() == 0
    assert torch.equal(padded[0, :, 505 + 3:505 + 14, 505 + 3:505 + 14], z[0][:, :-3, :-3])
    monkeypatch.setattr(G, "PATCH_SIZE", 14)   # small stand-in for the 1024-px conditioning image
    conds = G.cond_images_for_grid(z, geom, pos, fill_color=0.95)
    assert conds.shape == (4, 3, 14, 14)
    # patch (0,0) centre (4,4) moves to the image centre (7,7): shift +3, top/left 3 rows filled
    assert torch.equal(conds[0][:, 3:, 3:], z[0][:, :-3, :-3]) and (conds[0][:, :3] == 0.95).all()
    small = G.GridGeometry(8, 6, 2, 12, 28)
    patches = [torch.full((3, 16, 16), float(k)) for k in range(4)]
    full = G.stitch_canvas(patches, pos, small, background=None, patch_size=16)
    assert full.shape == (1, 3, 28, 28)
    assert (full[0, :, :12, :12] == 0).all() and (full[0, :, 12:, 12:] == 3).all()
    assert (full[0, :, 12:16, :12] == 2).all()  # later patch overwrites the overlap (index order)
    bg = torch.zeros(1, 3, 7, 7) + 0.5
    assert G.stitch_canvas(patches[:1], pos[:1], small, background=bg, patch_size=16)[0, :, 20:, 20:].eq(0.5).all()


def test_windowed_cond_images_equal_the_reference_roll_and_large_resize_equals_interpolate(monkeypatch):
    """cond_images_for_grid computes only the cropped window; the reference rolls the whole image, fills and crops
    (sample_ultra_res.py:356-391).  Restated naively here, including the shift == 0 quirk and odd crop margins."""
    monkeypatch.setattr(G, "PATCH_SIZE", 14)
    for W, pw, dist, n in ((51, 9, 6, 8), (40, 8, 4, 9), (14, 2, 3, 4)):
        z = torch.rand(1, 3, W, W, generator=torch.Generator().manual_seed(W))
        geom = G.GridGeometry(pw, dist, n, 10, 0)
        pos = geom.positions
        got = G.cond_images_for_grid(z, geom, pos, fill_color=0.95)
        for k, (i, j) in enumerate(pos):
            cy, cx = i * dist + pw // 2, j * dist + pw // 2
            sy, sx = W // 2 - cy, W // 2 - cx
            img = torch.roll(z[0], shifts=(sy, sx), dims=(1, 2))
            if sy > 0:
                img[:, :sy, :] = 0.95
            else:
                img[:, sy:, :] = 0.95
            if sx > 0:
                img[:, :, :sx] = 0.95
            else:
                img[:, :, sx:] = 0.95
            off = int(round((W - 14) / 2.0))
            assert torch.equal(got[k], img[:, off:off + 14, off:off + 14]), (W, i, j)
        assert any(W // 2 == i * dist + pw // 2 for i in range(n)) or W == 51   # the shift == 0 row is exercised
    # banded bilinear resize (the mag-2 canvas exceeds torch's 2^31-element device kernel) = F.interpolate
    g = torch.Generator().manual_seed(9)
    for hs, size in ((13, 83), (64, 64), (50, 37), (7, 400)):
        x = torch.rand(1, 3, hs, hs, generator=g)
        want = torch.nn.functional.interpolate(x, size=(size, size), mode="bilinear", align_corners=False)
        got = G.bilinear_resize_large(x, size, band_rows=29)
        assert torch.allclose(got, want, atol=1e-6, rtol=0), (hs, size, float((got - want).abs().max()))


def _stub_sample_fn(stage, tasks, lows, conds, ips, ims):
    """Deterministic stand-in for the sampler: a function of every input the real one consumes."""
    S = G.PATCH_SIZES[stage]
    outs = []
    for (c, i, j), low, cond, ip, im in zip(tasks, lows, conds, ips, ims):
        base = torch.full((3, S, S), 0.01 * (c + 1) + 0.1 * i + 0.001 * j + stage)
        if low is not None:
            base = base + torch.nn.functional.interpolate(low[None], S, mode="nearest")[0] * 0.5
        if cond is not None:
            base = base + cond[:3].mean() * 0.25
        base = torch.where(im.bool()[None], ip, base + 0.3 * ip.mean())
        outs.append(base)
    return outs


def _run_grid(world, rank=0, port=None, out=None):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos = [(i, j) for i in range(3) for j in range(3)]
        g = torch.Generator().manual_seed(0)
        conds = [torch.rand(9, 3, 8, 8, generator=g), torch.rand(9, 3, 8, 8, generator=g)]
        res = D.sample_grids(_stub_sample_fn, (1, 2), [pos, pos], conds, 0.25, [3, 3], orientations=[-1, 1])
        flat = torch.stack([torch.stack(r) for r in res])
        if out is not None:
            out[rank] = flat
        return flat
    finally:
        if world > 1:
            torch.distributed.destroy_process_group()


def _worker(rank, world, port, out):
    _run_grid(world, rank, port, out)


def test_two_rank_gloo_schedule_equals_single_process():
    single = _run_grid(1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.equal(out[0], single) and torch.equal(out[1], single)
    # stage 2 really consumed stage 1 and the neighbours: overlaps were pasted from finished patches
    assert single.shape == (2, 9, 3, 256, 256)


def _run_grid3(world, rank=0, port=None, out=None, pipeline=True, canvases=3):
    """3 canvases x 3 stages, 4x4 grids of mixed orientation - the shape of `bench.py --workload grid --canvases 3`."""
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 8, 2: 16, 3: 32})
    try:
        n = 4
        pos = [(i, j) for i in range(n) for j in range(n)]
        g = torch.Generator().manual_seed(1)
        conds = [torch.rand(n * n, 3, 8, 8, generator=g) for _ in range(canvases)]
        res = D.sample_grids(_stub_sample_fn, (1, 2, 3), [pos] * canvases, conds, 0.25, [n] * canvases,
                             orientations=[-1, 1, -1][:canvases], pipeline=pipeline)
        flat = torch.stack([torch.stack(r) for r in res])
        if out is not None:
            out[rank] = flat
        return flat
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
        if world > 1:
            torch.distributed.destroy_process_group()


def _worker3(rank, world, port, out):
    _run_grid3(world, rank, port, out)


def test_four_rank_gloo_three_canvases_pipelined_equals_single_process_with_stage_barriers():
    """world_size 4, --canvases 3, stages pipelined (a patch's stage s starts once its own stage s-1 and its
    neighbours' stage s are done) against ONE process running the reference's order (stage barrier,
    sample_ultra_res.py:264-270): bit-equal patches on every rank."""
    single = _run_grid3(1, pipeline=False)
    assert torch.equal(single, _run_grid3(1, pipeline=True))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_worker3, args=(4, port, out), nprocs=4, join=True)
    for r in range(4):
        assert torch.equal(out[r], single), r
    assert single.shape == (3, 16, 3, 32, 32)


def _run_grid8(world, rank=0, port=None, out=None, gather="all"):
    """One 8x8 canvas (the shape of BASELINE configs[4]) through three pipelined stages, plus a filtered 5x5 canvas of
    the other orientation whose missing neighbours fall back to crops of the conditioning image."""
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 8, 2: 16, 3: 32})
    try:
        pos8 = [(i, j) for i in range(8) for j in range(8)]
        pos5 = [(i, j) for i in range(5) for j in range(5) if (i, j) not in ((0, 0), (1, 3), (2, 2), (4, 0))]
        g = torch.Generator().manual_seed(2)
        conds = [torch.rand(len(pos8), 3, 40, 40, generator=g), torch.rand(len(pos5), 3, 40, 40, generator=g)]
        stats = {}
        res = D.sample_grids(_stub_sample_fn, (1, 2, 3), [pos8, pos5], conds, 0.25, [8, 5], orientations=[-1, 1],
                             patch_width=6, gather=gather, stats=stats)
        if out is not None:
            out[rank] = (res, stats)
        return res, stats
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
        if world > 1:
            torch.distributed.destroy_process_group()


def _worker8(rank, world, port, out, gather):
    _run_grid8(world, rank, port, out, gather)


@pytest.mark.parametrize("world,gather", [(8, "all"), (3, "none"), (2, "root")])
def test_strip_exchange_on_many_ranks_equals_single_process(world, gather):
    """The neighbour exchange moves only the overlap strips (sample_ultra_res.py:156-170), point to point, and the
    canvas is gathered once: 8 ranks (and 3, a world that does not divide the grid) give bit for bit the patches of
    one process; with gather="none" a rank ends with exactly the patches it sampled."""
    single, s1 = _run_grid8(1)
    assert s1["p2p_bytes_total"] == 0 and s1["blocking_collectives"] == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, port, out, gather), nprocs=world, join=True)
    held = [[0] * len(c) for c in single]
    for r in range(world):
        res, st = out[r]
        assert st["blocking_collectives"] == (0 if gather == "none" else 1)
        assert 0 < st["p2p_bytes_total"] < st["whole_patch_allgather_bytes_per_rank"]
        for c, canvas in enumerate(res):
            for k, p in enumerate(canvas):
                if gather == "all" or (gather == "root" and r == 0):
                    assert p is not None
                if p is not None:
                    assert torch.equal(p, single[c][k]), (r, c, k)
                    held[c][k] += 1
    want = {"all": (world, world), "none": (1, 1), "root": (1, 2)}[gather]   # root: rank 0 holds all, an owner its own
    assert all(want[0] <= h <= want[1] for row in held for h in row)
    assert sum(out[r][1]["p2p_bytes_sent_by_this_rank"] for r in range(world)) == out[0][1]["p2p_bytes_total"]


def _worker_coalesced(rank, world, port, out):
    """RCCL returns ONE work handle for a whole batch_isend_irecv call (a coalesced group), gloo one per operation:
    run the 2-rank grid with a batch_isend_irecv that hands back a single combined handle, as the RCCL path does."""
    import torch.distributed as dist

    real = dist.batch_isend_irecv

    class Combined:
        def __init__(self, works):
            self.works = works

        def wait(self):
            for w in self.works:
                w.wait()

    def coalesced(ops):
        return [Combined(real(ops))]

    D.dist.batch_isend_irecv = coalesced
    try:
        _run_grid8(world, rank, port, out, "all")
    finally:
        D.dist.batch_isend_irecv = real


def test_strip_exchange_with_one_handle_per_batch_as_rccl_returns_it():
    single, _ = _run_grid8(1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_worker_coalesced, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        res, st = out[r]
        assert st["blocking_collectives"] == 1 and st["p2p_bytes_total"] > 0
        assert all(torch.equal(p, q) for a, b in zip(res, single) for p, q in zip(a, b)), r


def test_exchange_plan_of_the_8x8_grid_moves_strips_not_patches():
    """BASELINE configs[4] on 8 ranks: what crosses xGMI before the final gather, from the plan alone."""
    pos = [(i, j) for i in range(8) for j in range(8)]
    cost = {1: 6.9, 2: 5.7, 3: 46.4}
    plan = D.ExchangePlan([pos], [-1], (1, 2, 3), 8, True, cost)
    # every task is dealt exactly once and depends only on earlier waves
    assert sorted(plan.owner) == sorted((s, 0, i, j) for s in (1, 2, 3) for i, j in pos)
    for t, need in plan.needs.items():
        assert all(plan.wave_of[p] < plan.wave_of[t] for _, p in need)
        assert len(need) <= 4 and [k for k, _ in need].count("low") == (0 if t[0] == 1 else 1)
    # bundle contents = exactly the needs whose producer sits on another rank
    remote = {(it, t) for t, need in plan.needs.items() for it in need if plan.owner[it[1]] != plan.owner[t]}
    bundled = [it for b in plan.bundles for items in b.values() for it in items]
    assert len(bundled) == len(remote) and set(bundled) == {it for it, _ in remote}
    for g, b in enumerate(plan.bundles):
        for (src, dst), items in b.items():
            assert src != dst and all(plan.owner[it[1]] == src and plan.wave_of[it[1]] == g for it in items)
    p2p = plan.p2p_bytes(0.25)
    whole = sum(4 * 3 * G.PATCH_SIZES[t[0]] ** 2 * 7 for w in plan.waves for t in w)   # every patch to 7 other ranks
    one_patch = 4 * 3 * 1024 * 1024
    print(f"8x8 grid on 8 ranks: p2p {p2p / 1e6:.0f} MB in {plan.p2p_messages()} messages; whole-patch all-gathers moved "
          f"{whole / 1e6:.0f} MB; final gather {64 * one_patch * 7 / 1e6:.0f} MB")
    assert p2p < 64 * 7.2e6 and p2p * 10 < whole
    # without a barrier between the waves the schedule is never longer than the per-wave maximum
    per_wave = sum(max(sum(cost[t[0]] for t in part) for part in parts) for parts in plan.parts)
    assert plan.makespan(cost) <= per_wave + 1e-9
    one = D.ExchangePlan([pos], [-1], (1, 2, 3), 1, True, cost)
    assert abs(one.makespan(cost) - 64 * sum(cost.values())) < 1e-6 and one.p2p_bytes(0.25) == 0


def test_stage_pipelining_and_the_deal_of_a_wave():
    n = 8
    pos = [(i, j) for i in range(n) for j in range(n)]
    # one canvas, three stages: 2n - 1 patch waves -> 2n + 1 generalised waves instead of 3 (2n - 1)
    gw = D.stage_waves([pos], [-1], (1, 2, 3))
    assert len(gw) == 2 * n - 1 + 2 and len(D.stage_waves([pos], [-1], (1, 2, 3), pipeline=False)) == 3 * (2 * n - 1)
    assert sorted(t for w in gw for t in w) == sorted((s, 0, i, j) for s in (1, 2, 3) for i, j in pos)
    seen = set()
    for w in gw:   # dependencies: previous stage of the same patch and same-stage neighbours are in EARLIER waves
        for (s, c, i, j) in w:
            assert s == 1 or (s - 1, c, i, j) in seen
            for nb in ((i - 1, j), (i, j - 1), (i - 1, j - 1)):
                assert nb[0] < 0 or nb[1] < 0 or (s, c) + nb in seen
        seen.update(w)
    # the single-stage slot bounds of SURVEY §8e are unchanged, and 3 canvases lift the 8-GPU bound past 6x
    waves = D.merged_waves([pos], [-1])
    assert [D.schedule_length(waves, g) for g in (1, 2, 4, 8)] == [64, 36, 22, 15]
    w3 = D.merged_waves([pos] * 3, [-1] * 3)
    assert 3 * 64 / D.schedule_length(w3, 8) > 6.0
    # the deal: deterministic, heaviest first onto the least-loaded rank (weighted by the stage's cost), a column
    # stays on its rank when the wave fits
    wave = gw[9]
    parts = D.assign_tasks(wave, 8)
    assert sorted(t for p in parts for t in p) == sorted(wave) and parts == D.assign_tasks(list(reversed(wave)), 8)
    cost = D.DEFAULT_STAGE_COST
    load = [sum(cost[t[0]] for t in p) for p in parts]
    assert max(load) <= sum(load) / 8 + max(cost.values()) and max(load) - min(load) <= max(cost.values())
    assert [sum(1 for t in p if t[0] == 3) for p in parts] == [1] * 8      # the 8 stage-3 patches: one per rank
    full = D.assign_tasks([(3, 0, i, 7 - i) for i in range(8)], 8)       # the longest anti-diagonal of stage 3
    assert all(len(p) == 1 and p[0][3] % 8 == r for r, p in enumerate(full))


def _np_rgb2hsv(arr):
    """skimage.color.rgb2hsv restated in numpy ((H,W,3) floats), the function the reference calls at
    sample_ultra_res.py:321."""
    import numpy as np

    out = np.empty_like(arr)
    v = arr.max(-1)
    delta = np.ptp(arr, -1)
    old = np.seterr(invalid="ignore", divide="ignore")
    s = delta / v
    s[delta == 0.0] = 0.0
    idx = arr[..., 0] == v
    out[idx, 0] = (arr[idx, 1] - arr[idx, 2]) / delta[idx]
    idx = arr[..., 1] == v
    out[idx, 0] = 2.0 + (arr[idx, 2] - arr[idx, 0]) / delta[idx]
    idx = arr[..., 2] == v
    out[idx, 0] = 4.0 + (arr[idx, 0] - arr[idx, 1]) / delta[idx]
    h = (out[..., 0] / 6.0) % 1.0
    h[delta == 0.0] = 0.0
    np.seterr(**old)
    return np.stack((h, np.nan_to_num(s), v), -1)


def _np_morph(m, k, op):
    """cv2.erode / cv2.dilate with a k x k ones kernel and the default border (outside pixels ignored)."""
    import numpy as np

    H, W = m.shape
    r = k // 2
    out = np.empty_like(m)
    for y in range(H):
        for x in range(W):
            win = m[max(0, y - r):y + r + 1, max(0, x - r):x + r + 1]
            out[y, x] = win.min() if op == "erode" else win.max()
    return out


@pytest.mark.parametrize("version", ["ultra", "airs"])
def test_tissue_mask_and_patch_filter_match_the_reference_recipe(version):
    import numpy as np

    g = torch.Generator().manual_seed(21)
    S = 96
    img = torch.rand(1, 3, S, S, generator=g) * 0.08 + 0.9          # near-white background
    img[0, :, 20:40, 30:70] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]   # a purple (hue 0.8) tissue blob
    img[0, :, 70:72, 5:7] = torch.tensor([0.7, 0.3, 0.8])[:, None, None]       # 2x2 speck: removed by the 5x5 erosion
    img[0, :, 50:60, 50:60] = 0.5                                              # grey square: delta == 0 -> hue 0
    img[0, :, 0, 0] = 0.0                                                      # black pixel: v == 0
    hsv = G.rgb_to_hsv(img[0])
    ref_hsv = _np_rgb2hsv(img[0].permute(1, 2, 0).numpy().astype(np.float32))
    assert np.allclose(hsv.permute(1, 2, 0).numpy(), ref_hsv, atol=1e-6)

    erode, dilate = 5, 11
    mask = G.tissue_mask(img, version=version, erode=erode, dilate=dilate)
    ref = ref_hsv[..., 2] > 0.1 if version == "airs" else np.logical_and(ref_hsv[..., 0] > 0.5, ref_hsv[..., 1] > 0.02)
    ref = _np_morph(_np_morph(ref.astype(np.uint8), erode, "erode"), dilate, "dilate")
    assert np.array_equal(mask.numpy(), ref > 0.5)
    if version == "ultra":
        assert mask[30, 50] and not mask[71, 6] and not mask[55, 55]

    geom = G.GridGeometry(patch_width=20, patch_dist=15, num_patches_width=7, out_patch_dist=768, canvas_width=0)
    got = G.tissue_patch_positions(mask, geom)
    want = [(i, j) for i in range(7) for j in range(7)
            if np.any(ref[i * 15:i * 15 + 20, j * 15:j * 15 + 20] > 0.5)]   # sample_ultra_res.py:343-352
    assert got == want
    if version == "ultra":
        assert 0 < len(got) < 49


def test_imagen_sample_fn_batches_a_wave_and_keeps_patch_order():
    """imagen_sample_fn(max_batch): one sample() call per chunk of a wave, every patch keeps its own
    conditioning / low-res / inpaint tensors, results come back in task order."""

    class FakeImagen:
        def __init__(self):
            self.calls = []

        def to(self, device):
            return self

        def sample(self, **kw):
            self.calls.append((kw["batch_size"], kw["start_at_unet_number"], kw["inpaint_resample_times"]))
            b = kw["batch_size"]
            assert kw["cond_images"].shape[0] == b and kw["inpaint_images"].shape[0] == b
            assert kw["inpaint_masks"].shape == (b, 4, 4) and kw["start_image_or_video"].shape[0] == b
            # encode which patch this is: mean of its cond image + 10 * mean of its low-res image
            tag = kw["cond_images"].mean(dim=(1, 2, 3)) + 10 * kw["start_image_or_video"].mean(dim=(1, 2, 3))
            return tag[:, None, None, None].expand(b, 3, 4, 4).clone()

    fake = FakeImagen()
    fn = D.imagen_sample_fn(lambda stage: fake, inpaint_resample=3, device=torch.device("cpu"), max_batch={2: 4, 3: 1})
    n = 6
    tasks = [(0, 0, k) for k in range(n)]
    conds = [torch.full((3, 4, 4), float(k)) for k in range(n)]
    lows = [torch.full((3, 2, 2), float(k) / 10) for k in range(n)]
    ips = [torch.zeros(3, 4, 4) for _ in range(n)]
    ims = [torch.zeros(4, 4) for _ in range(n)]
    out = fn(2, tasks, lows, conds, ips, ims)
    assert [c[0] for c in fake.calls] == [4, 2] and all(c[1:] == (2, 3) for c in fake.calls)
    assert [round(float(o.mean()), 4) for o in out] == [round(k + 10 * k / 10, 4) for k in range(n)]
    fake.calls.clear()
    out3 = fn(3, tasks[:3], lows[:3], conds[:3], ips[:3], ims[:3])
    assert [c[0] for c in fake.calls] == [1, 1, 1] and len(out3) == 3


def test_outpaint_canvas_equals_the_reference_loop_in_row_major_order():
    """outpainting.py:66-243 restated naively: patches in index order (row-major, which satisfies the
    above / left / above-left dependencies for orientation -1), inpaint tensors from whatever neighbour
    exists, no conditioning images, zero canvas.  The wavefront scheduler must give the same canvas."""
    sizes = {1: 8, 2: 16, 3: 32}
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update(sizes)
    n, overlap = 3, 0.25

    def stub(stage, task, low, ip, im):   # deterministic "sampler": depends on everything it is given
        S = sizes[stage]
        yy = torch.arange(S).float()[:, None] * 0.01 + torch.arange(S).float()[None, :] * 0.003
        base = (yy + 0.1 * stage + 0.07 * task[1] + 0.013 * task[2])[None].repeat(3, 1, 1)
        if low is not None:
            base = base + 0.5 * torch.nn.functional.interpolate(low[None], S, mode="nearest")[0]
        return torch.where(im.bool()[None], ip, base + 0.2 * ip.mean())

    def sample_fn(stage, tasks, lows, conds, ips, ims):
        assert all(c is None for c in conds)
        return [stub(stage, t, lo, ip, im) for t, lo, ip, im in zip(tasks, lows, ips, ims)]

    try:
        got = D.outpaint_canvas(sample_fn, n, overlap)[0]
        # the reference's loop
        pos = [(i, j) for i in range(n) for j in range(n)]
        prev = None
        for stage in (1, 2, 3):
            S, done = sizes[stage], {}
            ov = int(overlap * S)
            for idx, (i, j) in enumerate(pos):
                ip, im = torch.zeros(3, S, S), torch.zeros(S, S)
                a, l, al = done.get((i - 1, j)), done.get((i, j - 1)), done.get((i - 1, j - 1))
                if a is not None:
                    ip[:, :ov, :] = a[:, -ov:, :]
                    im[:ov, :] = 1
                if l is not None:
                    ip[:, :, :ov] = l[:, :, -ov:]
                    im[:, :ov] = 1
                if al is not None:
                    ip[:, :ov, :ov] = al[:, -ov:, -ov:]
                done[(i, j)] = stub(stage, (0, i, j), None if prev is None else prev[idx], ip, im)
            prev = [done[p] for p in pos]
        P = sizes[3]
        dist = int(P * (1 - overlap))
        want = torch.zeros(1, 3, P + (n - 1) * dist, P + (n - 1) * dist)
        for idx, (i, j) in enumerate(pos):
            want[0, :, i * dist:i * dist + P, j * dist:j * dist + P] = prev[idx]
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
    assert got.shape == want.shape and torch.equal(got, want)


def _worker_batches(rank, world, port, out):
    """Records every batch_isend_irecv call of the run: its wave-independent signature (the peers of its operations)."""
    import torch.distributed as dist

    real = dist.batch_isend_irecv
    calls = []

    def spy(ops):
        calls.append(sorted({op.peer for op in ops}))
        return real(ops)

    D.dist.batch_isend_irecv = spy
    try:
        _run_grid8(world, rank, port, None, "all")
    finally:
        D.dist.batch_isend_irecv = real
    out[rank] = calls


def test_exchange_posts_one_batch_per_peer_in_the_order_every_rank_shares():
    """a batch per (src, dst) pair, not one per wave - under RCCL a batch completes with its slowest
    peer.  Every call carries the operations of exactly one peer, and the calls of a rank follow the plan's
    (wave, low rank, high rank) order, which is what makes the order deadlock-free on one communicator."""
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_worker_batches, args=(world, port, out), nprocs=world, join=True)
    old = dict(G.PATCH_SIZES)
    G.PATCH_SIZES.update({1: 8, 2: 16, 3: 32})
    try:
        pos8 = [(i, j) for i in range(8) for j in range(8)]
        pos5 = [(i, j) for i in range(5) for j in range(5) if (i, j) not in ((0, 0), (1, 3), (2, 2), (4, 0))]
        plan = D.ExchangePlan([pos8, pos5], [-1, 1], (1, 2, 3), world, True, None)
    finally:
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
    for r in range(world):
        want = []
        for g in range(len(plan.waves)):
            pairs = sorted({(min(s_, d), max(s_, d)) for (s_, d) in plan.bundles[g] if r in (s_, d)})
            want.extend([[a if b == r else b] for a, b in pairs])
        assert out[r] == want, r
        assert all(len(c) == 1 for c in out[r])
