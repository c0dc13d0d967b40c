"""ORACLE — TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

CPU (stock PyTorch, fp32) restatement of the pieces of ``imagen-pytorch==1.18.5``
(reference ``requirements.txt:37``) that sit on the sampling hot path of
jameshball/kidney-diffusion: the Imagen-style ``Unet`` forward and the module tree /
state-dict key layout it implies.  The sampler lives in ``oracle/sampler_ref.py``.

Why "unpinned": the arithmetic lives in a third-party dependency whose source is NOT under
/root/reference and is not installed in this image; the reference ships no tests, seeds,
checkpoints or golden outputs.  This file follows the published
algorithm of that library as restated in SURVEY.md Appendix A, anchored on the reference's
own call sites:

  * constructor kwargs        train_ultra_res.py:27-60, train.py:28-65, train_uncond.py:28-61
  * Imagen(...) kwargs        train_ultra_res.py:79-90, train.py:83-93, train_uncond.py:79-90
  * sample(...) kwargs        sample_ultra_res.py:183-195, sample_cond.py:40-48,
                              sample_uncond.py:49-55
  * checkpoint layout         sample_ultra_res.py:53-63

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package (``kidney-diffusion_amd/imagen_pytorch``) never does.

Every op here is a plain ``torch``/``torch.nn.functional`` call, which is what the
reference's path reduces to when it runs on CPU.
"""
from __future__ import annotations

import math
import re
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- helpers
def exists(v):
    return v is not None


def default(v, d):
    if exists(v):
        return v
    return d() if callable(d) else d


def cast_tuple(v, length=None):
    if isinstance(v, list):
        v = tuple(v)
    out = v if isinstance(v, tuple) else ((v,) * default(length, 1))
    if exists(length):
        assert len(out) == length
    return out


def resize_image_to(image, target_image_size, mode="nearest"):
    """F.interpolate(mode='nearest'); identity when sizes match."""
    if image.shape[-1] == target_image_size:
        return image
    return F.interpolate(image, target_image_size, mode=mode)


# --------------------------------------------------------------------------- norms
class LayerNorm(nn.Module):
    """Gain-only layer norm over the last dim: (x-mu)*rsqrt(var+1e-5)*g   (SURVEY A.1)."""

    def __init__(self, feats):
        super().__init__()
        self.g = nn.Parameter(torch.ones(feats))

    def forward(self, x):
        var = torch.var(x, dim=-1, unbiased=False, keepdim=True)
        mean = torch.mean(x, dim=-1, keepdim=True)
        return (x - mean) * (var + 1e-5).rsqrt() * self.g


class Identity(nn.Module):
    def forward(self, x, *a, **k):
        return x


class Parallel(nn.Module):
    def __init__(self, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)

    def forward(self, x):
        return sum(fn(x) for fn in self.fns)


class PixelUnshuffle2(nn.Module):
    """einops 'b c (h s1) (w s2) -> b (c s1 s2) h w', s1=s2=2 — parameter-free."""

    def forward(self, x):
        b, c, h, w = x.shape
        x = x.reshape(b, c, h // 2, 2, w // 2, 2)
        return x.permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)


def Downsample(dim, dim_out=None, form="unshuffle"):
    """1.18.x: pixel-unshuffle + 1x1 conv (keys `<pre>.1.weight` [d_out, 4 d, 1, 1]).  Earlier library versions:
    a strided convolution `Conv2d(dim, dim_out, 4, 2, 1)` (key `<pre>.weight` [d_out, d, 4, 4]) - SURVEY A.1's
    version fork, selected by a checkpoint's key shapes (Unet._load_from_state_dict)."""
    dim_out = default(dim_out, dim)
    if form == "conv4x4":
        return nn.Conv2d(dim, dim_out, 4, 2, 1)
    assert form == "unshuffle"
    return nn.Sequential(PixelUnshuffle2(), nn.Conv2d(dim * 4, dim_out, 1))


class PixelShuffleUpsample(nn.Module):
    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = default(dim_out, dim)
        conv = nn.Conv2d(dim, dim_out * 4, 1)
        self.net = nn.Sequential(conv, nn.SiLU(), nn.PixelShuffle(2))
        o, i, h, w = conv.weight.shape
        cw = torch.empty(o // 4, i, h, w)
        nn.init.kaiming_uniform_(cw)
        conv.weight.data.copy_(cw.repeat_interleave(4, dim=0))
        nn.init.zeros_(conv.bias.data)

    def forward(self, x):
        return self.net(x)


class LearnedSinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))

    def forward(self, x):
        x = x[:, None]
        freqs = x * self.weights[None, :] * 2 * math.pi
        return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


class Unflatten2(nn.Module):
    """'b (r d) -> b r d' — parameter-free."""

    def __init__(self, r):
        super().__init__()
        self.r = r

    def forward(self, x):
        return x.reshape(x.shape[0], self.r, -1)


class CrossEmbedLayer(nn.Module):
    def __init__(self, dim_in, kernel_sizes, dim_out=None, stride=2):
        super().__init__()
        dim_out = default(dim_out, dim_in)
        kernel_sizes = sorted(kernel_sizes)
        n = len(kernel_sizes)
        dim_scales = [int(dim_out / (2 ** i)) for i in range(1, n)]
        dim_scales = [*dim_scales, dim_out - sum(dim_scales)]
        self.convs = nn.ModuleList([
            nn.Conv2d(dim_in, ds, k, stride=stride, padding=(k - stride) // 2)
            for k, ds in zip(kernel_sizes, dim_scales)
        ])

    def forward(self, x):
        return torch.cat([conv(x) for conv in self.convs], dim=1)


# --------------------------------------------------------------------------- attention
class _QKNorm:
    """Similarity variants (SURVEY A.1 marks this version-dependent; 1.18.5 itself cannot be inspected here):
    mode 0  sim = (q * dim_head^-0.5) . k                       library default, what the reference's configs get
    mode 1  `cosine_sim_attn=True` of 1.18.x: sim = l2norm(q) . l2norm(k) * 16      (q * self.scale with scale 1)
    mode 2  later versions: sim = (l2norm(q) * q_scale) . (l2norm(k) * k_scale) * 8, learned per-channel scales
    k includes the null key and the context keys (they are concatenated before the normalisation)."""
    qk_norm = 0

    def set_qk_norm(self, mode):
        self.qk_norm = mode
        if mode == 2 and not hasattr(self, "q_scale"):
            d = self.null_kv.shape[-1] if hasattr(self, "null_kv") else self.dim_head
            self.q_scale = nn.Parameter(torch.ones(d))
            self.k_scale = nn.Parameter(torch.ones(d))
        if mode != 2 and hasattr(self, "q_scale"):
            del self.q_scale, self.k_scale

    def similarity_inputs(self, q, k):
        if self.qk_norm == 0:
            return q * self.scale, k, 1.0
        q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
        if self.qk_norm == 1:
            return q, k, 16.0
        return q * self.q_scale, k * self.k_scale, 8.0


class Attention(nn.Module, _QKNorm):
    """Multi-query self attention (one shared k/v head) with learned null k/v and
    optional context k/v."""

    def __init__(self, dim, *, dim_head=64, heads=8, context_dim=None):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner = dim_head * heads
        self.norm = LayerNorm(dim)
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.to_context = (
            nn.Sequential(nn.LayerNorm(context_dim), nn.Linear(context_dim, dim_head * 2))
            if exists(context_dim) else None
        )
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), LayerNorm(dim))

    def forward(self, x, context=None):
        b, n, _ = x.shape
        h = self.heads
        x = self.norm(x)
        q = self.to_q(x)
        k, v = self.to_kv(x).chunk(2, dim=-1)
        q = q.reshape(b, n, h, -1).permute(0, 2, 1, 3)
        nk, nv = self.null_kv.unbind(dim=-2)
        k = torch.cat((nk.expand(b, 1, -1), k), dim=-2)
        v = torch.cat((nv.expand(b, 1, -1), v), dim=-2)
        if exists(context):
            assert exists(self.to_context)
            ck, cv = self.to_context(context).chunk(2, dim=-1)
            k = torch.cat((ck, k), dim=-2)
            v = torch.cat((cv, v), dim=-2)
        q, k, sim_scale = self.similarity_inputs(q, k)
        sim = torch.einsum("bhid,bjd->bhij", q, k) * sim_scale
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.einsum("bhij,bjd->bhid", attn, v)
        out = out.permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


class CrossAttention(nn.Module, _QKNorm):
    """Multi-head cross attention of feature tokens to the conditioning tokens."""

    def __init__(self, dim, *, context_dim=None, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner = dim_head * heads
        context_dim = default(context_dim, dim)
        self.norm = LayerNorm(dim)
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), LayerNorm(dim))

    def forward(self, x, context):
        b, n, _ = x.shape
        h = self.heads
        x = self.norm(x)
        q = self.to_q(x)
        k, v = self.to_kv(context).chunk(2, dim=-1)
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        nk, nv = self.null_kv.unbind(dim=-2)
        k = torch.cat((nk.expand(b, h, 1, -1), k), dim=-2)
        v = torch.cat((nv.expand(b, h, 1, -1), v), dim=-2)
        q, k, sim_scale = self.similarity_inputs(q, k)
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * sim_scale
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.einsum("bhij,bhjd->bhid", attn, v)
        out = out.permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


def FeedForward(dim, mult=2):
    hidden = int(dim * mult)
    return nn.Sequential(
        LayerNorm(dim),
        nn.Linear(dim, hidden, bias=False),
        nn.GELU(),
        LayerNorm(hidden),
        nn.Linear(hidden, dim, bias=False),
    )


class TransformerBlock(nn.Module):
    def __init__(self, dim, *, depth=1, heads=8, dim_head=32, ff_mult=2, context_dim=None):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([
                Attention(dim=dim, heads=heads, dim_head=dim_head, context_dim=context_dim),
                FeedForward(dim=dim, mult=ff_mult),
            ]) for _ in range(depth)
        ])

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
        for attn, ff in self.layers:
            x = attn(x, context=context) + x
            x = ff(x) + x
        return x.reshape(b, h, w, c).permute(0, 3, 1, 2)


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kw):
        return self.fn(x, **kw) + x


class ResidualAttentionBlock(nn.Module):
    """The mid-block attention of earlier library versions: `EinopsToAndFrom('b c h w', 'b (h w) c',
    Residual(Attention(mid_dim)))` - attention + residual WITHOUT the feed-forward of a TransformerBlock; state-dict
    keys `mid_attn.fn.fn.*` (SURVEY A.1's version fork)."""

    def __init__(self, dim, *, heads=8, dim_head=64):
        super().__init__()
        self.fn = _Residual(Attention(dim=dim, heads=heads, dim_head=dim_head))

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        x = self.fn(x.permute(0, 2, 3, 1).reshape(b, h * w, c))
        return x.reshape(b, h, w, c).permute(0, 3, 1, 2)


class PerceiverAttention(nn.Module, _QKNorm):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), nn.LayerNorm(dim))

    def forward(self, x, latents):
        x = self.norm(x)
        latents = self.norm_latents(latents)
        b, h = x.shape[0], self.heads
        q = self.to_q(latents)
        kv_input = torch.cat((x, latents), dim=-2)
        k, v = self.to_kv(kv_input).chunk(2, dim=-1)
        split = lambda t: t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        q, k, sim_scale = self.similarity_inputs(q, k)
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * sim_scale
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.einsum("bhij,bhjd->bhid", attn, v)
        out = out.permute(0, 2, 1, 3).reshape(b, q.shape[2], -1)
        return self.to_out(out)


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth, dim_head=64, heads=8, num_latents=64,
                 num_latents_mean_pooled=4, max_seq_len=512, ff_mult=4):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, dim)
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.to_latents_from_mean_pooled_seq = None
        if num_latents_mean_pooled > 0:
            self.to_latents_from_mean_pooled_seq = nn.Sequential(
                LayerNorm(dim),
                nn.Linear(dim, dim * num_latents_mean_pooled),
                Unflatten2(num_latents_mean_pooled),
            )
        self.layers = nn.ModuleList([
            nn.ModuleList([
                PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                FeedForward(dim=dim, mult=ff_mult),
            ]) for _ in range(depth)
        ])

    def forward(self, x):
        n = x.shape[1]
        pos_emb = self.pos_emb(torch.arange(n, device=x.device))
        x_with_pos = x + pos_emb
        latents = self.latents.expand(x.shape[0], -1, -1)
        if exists(self.to_latents_from_mean_pooled_seq):
            meanpooled = x.mean(dim=1)
            latents = torch.cat((self.to_latents_from_mean_pooled_seq(meanpooled), latents), dim=-2)
        for attn, ff in self.layers:
            latents = attn(x_with_pos, latents) + latents
            latents = ff(latents) + latents
        return latents


# --------------------------------------------------------------------------- res blocks
class GlobalContext(nn.Module):
    def __init__(self, *, dim_in, dim_out):
        super().__init__()
        self.to_k = nn.Conv2d(dim_in, 1, 1)
        hidden = max(3, dim_out // 2)
        self.net = nn.Sequential(
            nn.Conv2d(dim_in, hidden, 1), nn.SiLU(), nn.Conv2d(hidden, dim_out, 1), nn.Sigmoid()
        )

    def forward(self, x):
        b, c, h, w = x.shape
        context = self.to_k(x).reshape(b, 1, h * w)
        xf = x.reshape(b, c, h * w)
        out = torch.einsum("bin,bcn->bci", context.softmax(dim=-1), xf)
        return self.net(out.unsqueeze(-1))


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.groupnorm = nn.GroupNorm(groups, dim)
        self.activation = nn.SiLU()
        self.project = nn.Conv2d(dim, dim_out, 3, padding=1)

    def forward(self, x, scale_shift=None):
        x = self.groupnorm(x)
        if exists(scale_shift):
            scale, shift = scale_shift
            x = x * (scale + 1) + shift
        x = self.activation(x)
        return self.project(x)


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, *, cond_dim=None, time_cond_dim=None, groups=8,
                 use_gca=False, heads=8, dim_head=64):
        super().__init__()
        self.time_mlp = None
        if exists(time_cond_dim):
            self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_cond_dim, dim_out * 2))
        self.cross_attn = None
        if exists(cond_dim):
            self.cross_attn = CrossAttention(dim=dim_out, context_dim=cond_dim,
                                             heads=heads, dim_head=dim_head)
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.gca = GlobalContext(dim_in=dim_out, dim_out=dim_out) if use_gca else None
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else Identity()

    def forward(self, x, time_emb=None, cond=None):
        scale_shift = None
        if exists(self.time_mlp) and exists(time_emb):
            te = self.time_mlp(time_emb)[:, :, None, None]
            scale_shift = te.chunk(2, dim=1)
        h = self.block1(x)
        if exists(self.cross_attn):
            assert exists(cond)
            b, c, hh, ww = h.shape
            tok = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
            tok = self.cross_attn(tok, context=cond) + tok
            h = tok.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        h = self.block2(h, scale_shift=scale_shift)
        if exists(self.gca):
            h = h * self.gca(h)
        return h + self.res_conv(x)


# --------------------------------------------------------------------------- Unet
class Unet(nn.Module):
    """Restatement of imagen_pytorch.Unet for the kwargs the reference passes
    (train_ultra_res.py:29-60; train.py:30-65; train_uncond.py:30-61).  Defaults the
    reference never overrides are fixed at the library's values."""

    def __init__(
        self, *, dim, text_embed_dim=768, num_resnet_blocks=1, cond_dim=None,
        num_time_tokens=2, learned_sinu_pos_emb_dim=16, dim_mults=(1, 2, 4, 8),
        cond_images_channels=0, channels=3, channels_out=None, attn_dim_head=64,
        attn_heads=8, ff_mult=2.0, lowres_cond=False, layer_attns=True,
        layer_attns_depth=1, attend_at_middle=True, layer_cross_attns=True,
        cond_on_text=True, max_text_len=256, resnet_groups=8,
        init_cross_embed_kernel_sizes=(3, 7, 15), attn_pool_text=True,
        attn_pool_num_latents=32, memory_efficient=False,
        init_conv_to_final_conv_residual=False, use_global_context_attn=True,
        scale_skip_connection=True, final_conv_kernel_size=3, cosine_sim_attn=False, attn_qk_norm=None,
        downsample_form="unshuffle", mid_attn_form="transformer",
    ):
        super().__init__()
        self._locals = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        self.channels = channels
        self.channels_out = default(channels_out, channels)
        init_channels = channels * (1 + int(lowres_cond))
        init_dim = dim
        self.has_cond_image = cond_images_channels > 0
        self.cond_images_channels = cond_images_channels
        init_channels += cond_images_channels

        self.init_conv = CrossEmbedLayer(init_channels, dim_out=init_dim,
                                         kernel_sizes=init_cross_embed_kernel_sizes, stride=1)
        dims = [init_dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))

        cond_dim = default(cond_dim, dim)
        time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.cond_dim, self.time_cond_dim = cond_dim, time_cond_dim

        def time_trio():
            return (
                nn.Sequential(LearnedSinusoidalPosEmb(learned_sinu_pos_emb_dim),
                              nn.Linear(learned_sinu_pos_emb_dim + 1, time_cond_dim), nn.SiLU()),
                nn.Sequential(nn.Linear(time_cond_dim, time_cond_dim)),
                nn.Sequential(nn.Linear(time_cond_dim, cond_dim * num_time_tokens),
                              Unflatten2(num_time_tokens)),
            )

        self.to_time_hiddens, self.to_time_cond, self.to_time_tokens = time_trio()
        self.lowres_cond = lowres_cond
        if lowres_cond:
            (self.to_lowres_time_hiddens, self.to_lowres_time_cond,
             self.to_lowres_time_tokens) = time_trio()

        self.norm_cond = nn.LayerNorm(cond_dim)

        self.text_to_cond = None
        if cond_on_text:
            assert exists(text_embed_dim)
            self.text_to_cond = nn.Linear(text_embed_dim, cond_dim)
        self.cond_on_text = cond_on_text

        self.attn_pool = PerceiverResampler(
            dim=cond_dim, depth=2, dim_head=attn_dim_head, heads=attn_heads,
            num_latents=attn_pool_num_latents) if attn_pool_text else None

        self.max_text_len = max_text_len
        self.null_text_embed = nn.Parameter(torch.randn(1, max_text_len, cond_dim))
        self.null_text_hidden = nn.Parameter(torch.randn(1, time_cond_dim))

        self.to_text_non_attn_cond = None
        if cond_on_text:
            self.to_text_non_attn_cond = nn.Sequential(
                nn.LayerNorm(cond_dim), nn.Linear(cond_dim, time_cond_dim), nn.SiLU(),
                nn.Linear(time_cond_dim, time_cond_dim))

        attn_kwargs = dict(heads=attn_heads, dim_head=attn_dim_head)
        num_layers = len(in_out)
        num_resnet_blocks = cast_tuple(num_resnet_blocks, num_layers)
        resnet_groups = cast_tuple(resnet_groups, num_layers)
        layer_attns = cast_tuple(layer_attns, num_layers)
        layer_attns_depth = cast_tuple(layer_attns_depth, num_layers)
        layer_cross_attns = cast_tuple(layer_cross_attns, num_layers)
        resnet_klass = partial(ResnetBlock, **attn_kwargs)

        self.init_resnet_block = resnet_klass(
            init_dim, init_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[0],
            use_gca=use_global_context_attn) if memory_efficient else None

        self.skip_connect_scale = 1.0 if not scale_skip_connection else (2 ** -0.5)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        layer_params = [num_resnet_blocks, resnet_groups, layer_attns, layer_attns_depth,
                        layer_cross_attns]
        reversed_layer_params = [tuple(reversed(p)) for p in layer_params]
        skip_connect_dims = []

        for ind, ((dim_in, dim_out), n_blocks, groups, layer_attn, attn_depth,
                  layer_cross_attn) in enumerate(zip(in_out, *layer_params)):
            is_last = ind >= (num_layers - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:
                pre_downsample = Downsample(dim_in, dim_out, downsample_form)
                current_dim = dim_out
            skip_connect_dims.append(current_dim)
            post_downsample = None
            if not memory_efficient:
                post_downsample = Downsample(current_dim, dim_out, downsample_form) if not is_last else Parallel(
                    nn.Conv2d(dim_in, dim_out, 3, padding=1), nn.Conv2d(dim_in, dim_out, 1))
            attn = TransformerBlock(dim=current_dim, depth=attn_depth, ff_mult=ff_mult,
                                    context_dim=cond_dim, **attn_kwargs) if layer_attn else Identity()
            self.downs.append(nn.ModuleList([
                pre_downsample,
                resnet_klass(current_dim, current_dim, cond_dim=layer_cond_dim,
                             time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([
                    ResnetBlock(current_dim, current_dim, time_cond_dim=time_cond_dim,
                                groups=groups, use_gca=use_global_context_attn)
                    for _ in range(n_blocks)]),
                attn,
                post_downsample,
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim,
                                      time_cond_dim=time_cond_dim, groups=resnet_groups[-1])
        self.mid_attn = None
        if attend_at_middle:
            self.mid_attn = TransformerBlock(mid_dim, depth=1, **attn_kwargs) if mid_attn_form == "transformer" \
                else ResidualAttentionBlock(mid_dim, **attn_kwargs)
        assert mid_attn_form in ("transformer", "residual_attention")
        self.downsample_form, self.mid_attn_form = downsample_form, mid_attn_form
        self.mid_block2 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim,
                                      time_cond_dim=time_cond_dim, groups=resnet_groups[-1])

        for ind, ((dim_in, dim_out), n_blocks, groups, layer_attn, attn_depth,
                  layer_cross_attn) in enumerate(zip(reversed(in_out), *reversed_layer_params)):
            is_last = ind == (num_layers - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            skip_dim = skip_connect_dims.pop()
            attn = TransformerBlock(dim=dim_out, depth=attn_depth, ff_mult=ff_mult,
                                    context_dim=cond_dim, **attn_kwargs) if layer_attn else Identity()
            self.ups.append(nn.ModuleList([
                resnet_klass(dim_out + skip_dim, dim_out, cond_dim=layer_cond_dim,
                             time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([
                    ResnetBlock(dim_out + skip_dim, dim_out, time_cond_dim=time_cond_dim,
                                groups=groups, use_gca=use_global_context_attn)
                    for _ in range(n_blocks)]),
                attn,
                PixelShuffleUpsample(dim_out, dim_in) if (not is_last or memory_efficient) else Identity(),
            ]))

        self.init_conv_to_final_conv_residual = init_conv_to_final_conv_residual
        final_conv_dim = dim + (dim if init_conv_to_final_conv_residual else 0)
        self.final_res_block = ResnetBlock(final_conv_dim, dim, time_cond_dim=time_cond_dim,
                                           groups=resnet_groups[0], use_gca=True)
        final_conv_dim_in = dim + (channels if lowres_cond else 0)
        self.final_conv = nn.Conv2d(final_conv_dim_in, self.channels_out, final_conv_kernel_size,
                                    padding=final_conv_kernel_size // 2)
        nn.init.zeros_(self.final_conv.weight)
        nn.init.zeros_(self.final_conv.bias)
        self.attn_qk_norm = 0
        self.set_attn_qk_norm(int(attn_qk_norm) if exists(attn_qk_norm) else (1 if cosine_sim_attn else 0))

    # attention similarity switch (class _QKNorm): constructor kwarg, or taken from a checkpoint's key set
    def set_attn_qk_norm(self, mode):
        assert mode in (0, 1, 2)
        self.attn_qk_norm = mode
        self._locals["attn_qk_norm"] = mode
        for m in self.modules():
            if isinstance(m, _QKNorm):
                m.set_qk_norm(mode)

    def set_version_forks(self, downsample_form=None, mid_attn_form=None):
        """Rebuilds the modules of the two structural version forks in place (fresh parameters)."""
        L = self._locals
        dims = [L["dim"], *[L["dim"] * m for m in L["dim_mults"]]]
        if downsample_form is not None and downsample_form != self.downsample_form:
            for l, lvl in enumerate(self.downs):
                for slot in (0, 4):
                    if isinstance(lvl[slot], (nn.Sequential, nn.Conv2d)):
                        lvl[slot] = Downsample(dims[l], dims[l + 1], downsample_form)   # pre (mem-eff) and post alike
            self.downsample_form = L["downsample_form"] = downsample_form
        if mid_attn_form is not None and mid_attn_form != self.mid_attn_form and exists(self.mid_attn):
            kw = dict(heads=L["attn_heads"], dim_head=L["attn_dim_head"])
            self.mid_attn = TransformerBlock(dims[-1], depth=1, **kw) if mid_attn_form == "transformer" \
                else ResidualAttentionBlock(dims[-1], **kw)
            self.mid_attn_form = L["mid_attn_form"] = mid_attn_form
            self.set_attn_qk_norm(self.attn_qk_norm)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        conv4 = any(re.fullmatch(re.escape(prefix) + r"downs\.\d+\.[04]\.weight", k) and v.dim() == 4 and v.shape[-1] == 4
                    for k, v in state_dict.items())
        mine = any(k.startswith(prefix + "downs.") for k in state_dict)
        if mine:
            self.set_version_forks(
                downsample_form="conv4x4" if conv4 else "unshuffle",
                mid_attn_form="residual_attention" if any(k.startswith(prefix + "mid_attn.fn.") for k in state_dict)
                else "transformer")
        has = any(k.startswith(prefix) and k.endswith(".q_scale") for k in state_dict)
        if has and self.attn_qk_norm != 2:
            self.set_attn_qk_norm(2)
        elif not has and self.attn_qk_norm == 2 and any(k.startswith(prefix) for k in state_dict):
            self.set_attn_qk_norm(0)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # Imagen re-creates each Unet with the cascade-dependent kwargs
    def cast_model_parameters(self, *, lowres_cond, text_embed_dim, channels, channels_out,
                              cond_on_text):
        if (lowres_cond == self.lowres_cond and channels == self.channels
                and cond_on_text == self.cond_on_text
                and text_embed_dim == self._locals["text_embed_dim"]
                and channels_out == self.channels_out):
            return self
        updated = dict(lowres_cond=lowres_cond, text_embed_dim=text_embed_dim, channels=channels,
                       channels_out=channels_out, cond_on_text=cond_on_text)
        return self.__class__(**{**self._locals, **updated})

    def forward_with_cond_scale(self, *args, cond_scale=1.0, **kwargs):
        logits = self.forward(*args, **kwargs)
        if cond_scale == 1:
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1.0, **kwargs)
        return null_logits + (logits - null_logits) * cond_scale

    def text_conditioning(self, text_embeds, text_mask, cond_drop_prob=0.0):
        """(text_tokens, text_hiddens) — step-invariant part of the conditioning."""
        b = text_embeds.shape[0]
        keep = torch.full((b,), cond_drop_prob < 1.0, dtype=torch.bool, device=text_embeds.device)
        assert cond_drop_prob in (0.0, 1.0), "sampling only uses keep-all / drop-all"
        keep_embed, keep_hidden = keep[:, None, None], keep[:, None]
        text_tokens = self.text_to_cond(text_embeds)[:, : self.max_text_len]
        if exists(text_mask):
            text_mask = text_mask[:, : self.max_text_len]
        remainder = self.max_text_len - text_tokens.shape[1]
        if remainder > 0:
            text_tokens = F.pad(text_tokens, (0, 0, 0, remainder))
        if exists(text_mask):
            if remainder > 0:
                text_mask = F.pad(text_mask, (0, remainder), value=False)
            keep_embed = text_mask[:, :, None] & keep_embed
        text_tokens = torch.where(keep_embed, text_tokens, self.null_text_embed)
        if exists(self.attn_pool):
            text_tokens = self.attn_pool(text_tokens)
        text_hiddens = self.to_text_non_attn_cond(text_tokens.mean(dim=-2))
        text_hiddens = torch.where(keep_hidden, text_hiddens, self.null_text_hidden)
        return text_tokens, text_hiddens

    def forward(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None,
                text_embeds=None, text_mask=None, cond_images=None, cond_drop_prob=0.0):
        assert not (self.lowres_cond and not exists(lowres_cond_img))
        assert not (self.lowres_cond and not exists(lowres_noise_times))
        if exists(lowres_cond_img):
            x = torch.cat((x, lowres_cond_img), dim=1)
        assert not (self.has_cond_image ^ exists(cond_images))
        if exists(cond_images):
            assert cond_images.shape[1] == self.cond_images_channels
            cond_images = resize_image_to(cond_images, x.shape[-1])
            x = torch.cat((cond_images, x), dim=1)

        if getattr(self, "channels_last", False):
            # (test infrastructure: the same arithmetic through oneDNN's NHWC convolutions - a third less host time in the
            # full-size GPU tests, set there by helpers.fast_oracle; results differ from the NCHW run by the ~1e-6 rel-L2 of
            # a different fp32 summation order, which is the oracle's own resolution)
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.init_conv(x)
        if self.init_conv_to_final_conv_residual:
            init_conv_residual = x.clone()

        time_hiddens = self.to_time_hiddens(time)
        time_tokens = self.to_time_tokens(time_hiddens)
        t = self.to_time_cond(time_hiddens)
        if self.lowres_cond:
            lth = self.to_lowres_time_hiddens(lowres_noise_times)
            t = t + self.to_lowres_time_cond(lth)
            time_tokens = torch.cat((time_tokens, self.to_lowres_time_tokens(lth)), dim=-2)

        text_tokens = None
        if exists(text_embeds) and self.cond_on_text:
            text_tokens, text_hiddens = self.text_conditioning(text_embeds, text_mask, cond_drop_prob)
            t = t + text_hiddens

        c = time_tokens if not exists(text_tokens) else torch.cat((time_tokens, text_tokens), dim=-2)
        c = self.norm_cond(c)

        if exists(self.init_resnet_block):
            x = self.init_resnet_block(x, t)

        hiddens = []
        for pre_down, init_block, resnet_blocks, attn_block, post_down in self.downs:
            if exists(pre_down):
                x = pre_down(x)
            x = init_block(x, t, c)
            for rb in resnet_blocks:
                x = rb(x, t)
                hiddens.append(x)
            x = attn_block(x, c)
            hiddens.append(x)
            if exists(post_down):
                x = post_down(x)

        x = self.mid_block1(x, t, c)
        if exists(self.mid_attn):
            x = self.mid_attn(x)
        x = self.mid_block2(x, t, c)

        add_skip = lambda x: torch.cat((x, hiddens.pop() * self.skip_connect_scale), dim=1)
        for init_block, resnet_blocks, attn_block, upsample in self.ups:
            x = add_skip(x)
            x = init_block(x, t, c)
            for rb in resnet_blocks:
                x = add_skip(x)
                x = rb(x, t)
            x = attn_block(x, c)
            x = upsample(x)

        if self.init_conv_to_final_conv_residual:
            x = torch.cat((x, init_conv_residual), dim=1)
        x = self.final_res_block(x, t)
        if exists(lowres_cond_img):
            x = torch.cat((x, lowres_cond_img), dim=1)
        return self.final_conv(x)


class NullUnet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.lowres_cond = False
        self.dummy_parameter = nn.Parameter(torch.tensor([0.0]))

    def cast_model_parameters(self, *_, **__):
        return self

    def forward(self, x, *args, **kwargs):
        return x


def reinit_for_benchmark(unet: nn.Module, seed: int = 0, final_std: float = 0.02):
    """the library zero-inits ``final_conv``; synthetic benchmarks re-init it
    N(0, 0.02) so that outputs are non-trivial.  Deterministic under ``seed``."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        unet.final_conv.weight.copy_(
            torch.randn(unet.final_conv.weight.shape, generator=g) * final_std)
        unet.final_conv.bias.copy_(torch.randn(unet.final_conv.bias.shape, generator=g) * final_std)
    return unet
