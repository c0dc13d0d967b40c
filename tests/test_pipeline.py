"""One magnification level of the ultra-res pipeline (ultra_res/pipeline.py) against a naive
restatement of the reference's `generate_high_res_image` / `get_cond_images`
(sample_ultra_res.py:304-448), at a reduced patch size, with a deterministic stub sampler."""
import math

import pytest
import torch
import torch.nn.functional as F

from ultra_res import grid as G
from ultra_res import pipeline as P


@pytest.fixture
def small_patches(monkeypatch):
    monkeypatch.setattr(G, "PATCH_SIZE", 32)
    monkeypatch.setattr(G, "PATCH_SIZES", {1: 8, 2: 16, 3: 32})
    return 32


def _stub(stage, task, low, cond, ip, im):
    S = G.PATCH_SIZES[stage]
    base = torch.full((3, S, S), 0.05 * stage + 0.01 * task[1] + 0.002 * task[2])
    if low is not None:
        base = base + 0.5 * F.interpolate(low[None, :3], S, mode="nearest")[0]
    if cond is not None:
        base = base + 0.25 * F.interpolate(cond[None, :3], S, mode="nearest")[0]
    return torch.where(im.bool()[None], ip, base + 0.2 * ip.mean())


def _sample_fn(stage, tasks, lows, conds, ips, ims):
    return [_stub(stage, t, lo, c, ip, im) for t, lo, c, ip, im in zip(tasks, lows, conds, ips, ims)]


def _reference_level(zoomed, mag_level, overlap, pos_filter=None):
    """sample_ultra_res.py:304-448 written out naively (orientation chosen as there, patches processed
    in an order that respects the dependencies, fallback crops for filtered-out neighbours)."""
    PS = G.PATCH_SIZE
    W = zoomed.shape[3]
    pw = int(G.MAG_LEVEL_SIZES[mag_level] * PS / G.MAG_LEVEL_SIZES[mag_level - 1])
    dist = int(pw * (1 - overlap))
    n = 1 + math.ceil((W - pw) / dist)
    pos = [(i, j) for i in range(n) for j in range(n)]
    if pos_filter is not None:
        pos = [p for p in pos if pos_filter(p)]
    conds = []
    for i, j in pos:
        cy, cx = i * dist + pw // 2, j * dist + pw // 2
        sy, sx = W // 2 - cy, W // 2 - cx
        img = torch.roll(zoomed[0], shifts=(sy, sx), dims=(1, 2))
        if sy > 0:
            img[:, :sy, :] = 0.95
        else:
            img[:, sy:, :] = 0.95
        if sx > 0:
            img[:, :, :sx] = 0.95
        else:
            img[:, :, sx:] = 0.95
        assert W >= PS   # (narrower images are zero-padded by torchvision: covered in test_grid / test_geometry_pins)
        off = int(round((W - PS) / 2.0))   # torchvision CenterCrop (sample_ultra_res.py:391), half-to-even
        conds.append(img[:, off:off + PS, off:off + PS])

    def ready(p, o, rest):
        return (p[0] - 1, p[1]) not in rest and (p[0], p[1] + o) not in rest and (p[0] - 1, p[1] + o) not in rest

    tl = sum(ready(p, -1, set(pos)) for p in pos)
    tr = sum(ready(p, 1, set(pos)) for p in pos)
    o = -1 if tl > tr else 1
    prev = None
    for stage in (1, 2, 3):
        S = G.PATCH_SIZES[stage]
        ov = int(overlap * S)
        done, rest = {}, list(pos)
        while rest:
            p = next(q for q in rest if ready(q, o, set(rest)))
            rest.remove(p)
            i, j = p
            cond = conds[pos.index(p)]

            def nb(q, has_space, dy, dx):
                if q in pos:
                    return done[q]
                if not has_space:
                    return None
                ty = cond.shape[1] // 2 - pw // 2 + dy * dist
                tx = cond.shape[2] // 2 - pw // 2 + dx * dist
                return F.interpolate(cond[:3, ty:ty + pw, tx:tx + pw][None], size=(S, S), mode="bilinear",
                                     align_corners=False)[0]

            sa = i != 0
            sn = (o == 1 and j < n - 1) or (o == -1 and j > 0)
            a, nx, an = nb((i - 1, j), sa, -1, 0), nb((i, j + o), sn, 0, o), nb((i - 1, j + o), sa and sn, -1, o)
            ip, im = torch.zeros(3, S, S), torch.zeros(S, S)
            if a is not None:
                ip[:, :ov, :] = a[:, -ov:, :]
                im[:ov, :] = 1
            if nx is not None:
                if o == -1:
                    ip[:, :, :ov] = nx[:, :, -ov:]
                    im[:, :ov] = 1
                else:
                    ip[:, :, -ov:] = nx[:, :, :ov]
                    im[:, -ov:] = 1
            if an is not None:
                if o == -1:
                    ip[:, :ov, :ov] = an[:, -ov:, -ov:]
                else:
                    ip[:, :ov, -ov:] = an[:, -ov:, :ov]
            done[p] = _stub(stage, (0, i, j), None if prev is None else prev[pos.index(p)], cond, ip, im)
        prev = [done[p] for p in pos]
    od = int(PS * (1 - overlap))
    width = PS + (n - 1) * od
    full = F.interpolate(zoomed, size=(width, width), mode="bilinear", align_corners=False)
    for idx, (i, j) in enumerate(pos):
        full[0, :, i * od:i * od + PS, j * od:j * od + PS] = prev[idx]
    return full, pos


def test_mag1_level_equals_the_reference_driver(small_patches):
    zoomed = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    got = P.generate_high_res_image(_sample_fn, zoomed.clone(), 1, overlap=0.25)
    want, pos = _reference_level(zoomed.clone(), 1, 0.25)
    assert len(pos) == 100 and got.shape == want.shape == (1, 3, 32 + 9 * 24, 32 + 9 * 24)
    assert torch.equal(got, want)


def test_filtered_grid_uses_fallback_crops_and_keeps_the_background(small_patches):
    zoomed = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    keep = lambda p: (p[0] + 2 * p[1]) % 5 != 0 and p != (3, 3)   # holes inside the grid and on its border
    full_pos = [(i, j) for i in range(10) for j in range(10)]
    got = P.generate_high_res_image(_sample_fn, zoomed.clone(), 1, overlap=0.25,
                                    patch_pos=[p for p in full_pos if keep(p)])
    want, pos = _reference_level(zoomed.clone(), 1, 0.25, pos_filter=keep)
    assert 0 < len(pos) < 100
    assert torch.equal(got, want)


def test_mag2_level_filters_by_tissue_and_handles_an_empty_canvas(small_patches):
    white = torch.full((1, 3, 64, 64), 0.97)
    geom, pos = P.level_patches(white, 2, 0.25)
    assert pos == [] and geom.num_patches_width > 1
    out = P.generate_high_res_image(_sample_fn, white, 2, overlap=0.25)
    assert out.shape[-1] == geom.canvas_width and torch.allclose(out, torch.full_like(out, 0.97))
    img = white.clone()
    img[0, :, 10:20, 30:44] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]      # purple tissue
    geom, pos = P.level_patches(img, 2, 0.25)
    assert 0 < len(pos) < geom.num_patches_width ** 2
    out = P.generate_high_res_image(_sample_fn, img, 2, overlap=0.25)
    assert out.shape == (1, 3, geom.canvas_width, geom.canvas_width) and torch.isfinite(out).all()


def test_three_level_chain_is_mag0_then_two_grid_levels(small_patches):
    """sample_ultra_res.py:463-469: mag 0 is ONE unconditional sample without position or inpainting tensors
    (:88-91), each further level is generate_high_res_image over the previous level's image."""
    calls = []

    def fn0(stage, tasks, lows, conds, ips, ims):
        calls.append((stage, tasks, conds, ips, ims))
        S = G.PATCH_SIZES[stage]
        base = torch.linspace(0, 1, S * S).reshape(1, S, S).repeat(3, 1, 1) * (0.5 + 0.1 * stage)
        if lows[0] is not None:
            base = base + 0.3 * F.interpolate(lows[0][None], S, mode="nearest")[0]
        return [base]

    def fn1(stage, tasks, lows, conds, ips, ims):
        outs = _sample_fn(stage, tasks, lows, conds, ips, ims)
        for t, o in zip(tasks, outs):   # a few purple patches, so that the mag-2 tissue filter finds something
            if (t[1] + t[2]) % 6 == 0:
                o[:] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]
        return outs

    keep = lambda level, pos: pos if level == 1 else pos[:7]
    mag0, mag1, mag2 = P.generate_all_levels({0: fn0, 1: fn1, 2: _sample_fn}, overlap=0.25, patch_filter=keep)
    assert [c[0] for c in calls] == [1, 2, 3] and all(c[1] == [(0, 0, 0)] and c[2] == [None] and c[3] == [None] and c[4] == [None]
                                                       for c in calls)
    assert mag0.shape == (1, 3, 32, 32)
    want1 = P.generate_high_res_image(fn1, mag0.clone(), 1, overlap=0.25)
    assert torch.equal(mag1, want1) and mag1.shape[-1] == 32 + 9 * 24
    geom2, pos2 = P.level_patches(mag1, 2, 0.25)
    assert len(pos2) > 7
    want2 = P.generate_high_res_image(_sample_fn, mag1.clone(), 2, overlap=0.25, patch_pos=pos2[:7])
    assert torch.equal(mag2, want2) and mag2.shape[-1] == geom2.canvas_width


def _chain_fns():
    def fn0(stage, tasks, lows, conds, ips, ims):
        S = G.PATCH_SIZES[stage]
        base = torch.linspace(0, 1, S * S).reshape(1, S, S).repeat(3, 1, 1) * (0.5 + 0.1 * stage)
        if lows[0] is not None:
            base = base + 0.3 * F.interpolate(lows[0][None], S, mode="nearest")[0]
        return [base]

    def fn1(stage, tasks, lows, conds, ips, ims):
        outs = _sample_fn(stage, tasks, lows, conds, ips, ims)
        for t, o in zip(tasks, outs):
            if (t[1] + t[2]) % 6 == 0:
                o[:] = torch.tensor([0.75, 0.35, 0.8])[:, None, None]
        return outs

    return {0: fn0, 1: fn1, 2: _sample_fn}


def _chain(world, rank=0, port=None, out=None):
    import os

    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    old_ps, old = G.PATCH_SIZE, dict(G.PATCH_SIZES)
    G.PATCH_SIZE = 32
    G.PATCH_SIZES.update({1: 8, 2: 16, 3: 32})
    try:
        fns = _chain_fns()
        if world > 1 and rank != 0:   # only rank 0 samples the mag-0 image: the others must get it by broadcast
            def never(*a, **k):
                raise AssertionError("mag 0 is sampled on rank 0 alone")
            fns[0] = never
        keep = lambda level, pos: pos if level == 1 else pos[:7]
        res = P.generate_all_levels(fns, overlap=0.25, patch_filter=keep, device=torch.device("cpu"))
        if out is not None:
            out[rank] = res
        return res
    finally:
        G.PATCH_SIZE = old_ps
        G.PATCH_SIZES.clear()
        G.PATCH_SIZES.update(old)
        if world > 1:
            torch.distributed.destroy_process_group()


def _chain_worker(rank, world, port, out):
    _chain(world, rank, port, out)


def test_three_level_chain_on_two_gloo_ranks_equals_single_process():
    """`generate_all_levels` under torch.distributed: rank 0 samples the mag-0 image, every rank receives it
    into a buffer of the backend's kind (host memory under gloo, the rank's device under RCCL) and the two grid levels
    run sharded; both ranks end with the images of one process, bit for bit."""
    import socket

    import torch.multiprocessing as mp

    single = _chain(1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_chain_worker, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        for got, want in zip(out[r], single):
            assert got.shape == want.shape and torch.equal(got, want), r
