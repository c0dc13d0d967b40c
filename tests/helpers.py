"""Shared builders for the tests: matched (oracle, product) models with identical weights.

Reference kwargs being mirrored (at reduced `dim`, so the oracle finishes in seconds):
  ultra-res unet1/2/3   train_ultra_res.py:29-60
  uncond unet1          train_uncond.py:30-36
"""
import torch

from oracle import imagen_ref as R
from oracle import sampler_ref as RS

# name -> (Unet kwargs, lowres_cond)
UNET_KW = {
    "ultra1": dict(dim=32, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                   layer_cross_attns=(False, True, True, True), cond_images_channels=3),
    "ultra2": dict(dim=32, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                   layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                   init_conv_to_final_conv_residual=True, cond_images_channels=3),
    "ultra3": dict(dim=32, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True,
                   layer_attns=False, layer_cross_attns=(False, False, False, True),
                   init_conv_to_final_conv_residual=True, cond_images_channels=3),
    "uncond1": dict(dim=32, dim_mults=(1, 2, 4, 8), cond_dim=64, num_resnet_blocks=3,
                    layer_attns=(False, True, True, True), layer_cross_attns=(False, True, True, True)),
    "small1": dict(dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True),
                   layer_cross_attns=(False, True)),
    "small2": dict(dim=32, dim_mults=(1, 2), num_resnet_blocks=1, memory_efficient=True, layer_attns=(False, True),
                   layer_cross_attns=(False, True), init_conv_to_final_conv_residual=True, cond_images_channels=3),
}


def randomize_(module, seed, std=0.05):
    """Deterministic non-trivial weights: defaults under the seed, then every zero-/one-initialised
    tensor (final conv, norm gains/biases, pixel-shuffle biases) gets noise so no term is hidden."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if p.numel() == 1 and name.endswith("dummy_parameter"):
                continue
            if name.endswith(".g") or name.endswith("groupnorm.weight") or name.endswith("norm.weight") \
                    or name.endswith("norm_cond.weight") or name.endswith("norm_latents.weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "final_conv" in name:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel() if p.dim() > 1 else p.numel()
                p.copy_(torch.randn(p.shape, generator=g) * (fan_in ** -0.5))
    return module


def oracle_unet(name, lowres_cond=False, seed=0):
    kw = dict(UNET_KW[name])
    u = R.Unet(**kw, lowres_cond=lowres_cond, cond_on_text=False, text_embed_dim=None)
    return randomize_(u, seed)


def product_unet_like(oracle_u):
    import imagen_pytorch as ip

    kw = {k: v for k, v in oracle_u._locals.items()}
    u = ip.Unet(**kw)
    missing, unexpected = u.load_state_dict(oracle_u.state_dict(), strict=True)
    assert not missing and not unexpected
    return u


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def fast_oracle(u):
    """The oracle UNet `u` with its convolutions in channels_last (oneDNN's NHWC kernels: about a third less host time at full
    size; same arithmetic, another fp32 summation order - 1e-6 rel-L2, the oracle's own resolution).  Used by the full-size
    GPU tests only; the golden fixtures and the CPU tests run the default layout."""
    import torch

    u = u.to(memory_format=torch.channels_last)
    u.channels_last = True
    return u
