"""ORACLE — TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED** (see imagen_ref.py header).

CPU restatement of imagen-pytorch 1.18.5's ``GaussianDiffusionContinuousTimes`` and
``Imagen.sample / p_sample_loop / p_sample / p_mean_variance`` (SURVEY.md §3.2, Appendix
A.2) as the reference drives them from ``sample_ultra_res.py:183-195``,
``outpainting.py:146-157``, ``sample_cond.py:40-48`` and ``sample_uncond.py:49-55``.

The reference is unseeded; here every Gaussian draw goes through an injected
``noise_fn(tag, shape)`` so the HIP engine and this oracle can be fed identical noise.  Tags,
in the library's RNG draw order (SURVEY A.2 "RNG draw order"):

    ("lowres", stage)                      low-res conditioning augmentation noise
    ("init", stage)                        x_T
    ("inpaint", stage, k, r)               q_sample of the known pixels   (inpainting only)
    ("step", stage, k, r)                  p_sample noise (drawn on every step, also the last)
    ("renoise", stage, k, r)               q_sample_from_to re-noise      (inpainting only)

k = timestep index 0..T-1, r = resample index counting DOWN from resample_times-1 to 0.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .imagen_ref import NullUnet, Unet, cast_tuple, default, exists, resize_image_to


# --------------------------------------------------------------------------- schedules
def _log(t, eps=1e-12):
    return torch.log(t.clamp(min=eps))


def beta_linear_log_snr(t):
    return -torch.log(torch.expm1(1e-4 + 10 * (t ** 2)))


def alpha_cosine_log_snr(t, s: float = 0.008):
    return -_log((torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** -2) - 1, eps=1e-5)


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def _pad(x, t):
    return t.reshape(t.shape + (1,) * (x.ndim - t.ndim))


class GaussianDiffusionContinuousTimes:
    def __init__(self, *, noise_schedule, timesteps=1000):
        self.log_snr = {"linear": beta_linear_log_snr, "cosine": alpha_cosine_log_snr}[noise_schedule]
        self.num_timesteps = timesteps

    def get_times(self, batch, noise_level):
        return torch.full((batch,), noise_level, dtype=torch.float32)

    def get_sampling_timesteps(self, batch):
        times = torch.linspace(1.0, 0.0, self.num_timesteps + 1)
        times = times[None, :].expand(batch, -1)
        return list(zip(times[:, :-1].unbind(dim=-1), times[:, 1:].unbind(dim=-1)))

    def q_posterior(self, x_start, x_t, t, *, t_next):
        log_snr, log_snr_next = _pad(x_t, self.log_snr(t)), _pad(x_t, self.log_snr(t_next))
        alpha, sigma = log_snr_to_alpha_sigma(log_snr)
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next)
        c = -torch.expm1(log_snr - log_snr_next)
        mean = alpha_next * (x_t * (1 - c) / alpha + c * x_start)
        var = (sigma_next ** 2) * c
        return mean, var, _log(var, eps=1e-20)

    def q_sample(self, x_start, t, noise):
        alpha, sigma = log_snr_to_alpha_sigma(_pad(x_start, self.log_snr(t)))
        return alpha * x_start + sigma * noise

    def q_sample_from_to(self, x_from, from_t, to_t, noise):
        alpha, sigma = log_snr_to_alpha_sigma(_pad(x_from, self.log_snr(from_t)))
        alpha_to, sigma_to = log_snr_to_alpha_sigma(_pad(x_from, self.log_snr(to_t)))
        return x_from * (alpha_to / alpha) + noise * (sigma_to * alpha - sigma * alpha_to) / alpha

    def predict_start_from_v(self, x_t, t, v):
        alpha, sigma = log_snr_to_alpha_sigma(_pad(x_t, self.log_snr(t)))
        return alpha * x_t - sigma * v

    def predict_start_from_noise(self, x_t, t, noise):
        alpha, sigma = log_snr_to_alpha_sigma(_pad(x_t, self.log_snr(t)))
        return (x_t - sigma * noise) / alpha.clamp(min=1e-8)


def generator_noise_fn(seed: int):
    """Noise keyed by tag: every tag gets its own torch.Generator stream, so the HIP side
    can be handed exactly the same tensors in any order."""

    def fn(tag, shape):
        h = seed
        for part in tag:
            v = part if isinstance(part, int) else sum(ord(ch) * (i + 1) for i, ch in enumerate(part))
            h = (h * 1000003 + v + 0x9E3779B9) % (2 ** 63 - 1)
        return torch.randn(shape, generator=torch.Generator().manual_seed(h))

    return fn


# --------------------------------------------------------------------------- Imagen
class Imagen(nn.Module):
    """kwargs: train_ultra_res.py:79-90, train.py:83-93, train_uncond.py:79-90."""

    def __init__(self, unets, *, image_sizes, text_embed_dim=None, channels=3, timesteps=1000,
                 noise_schedules="cosine", pred_objectives="noise", random_crop_sizes=None,
                 lowres_noise_schedule="linear", lowres_sample_noise_level=0.2,
                 condition_on_text=True, dynamic_thresholding=True,
                 dynamic_thresholding_percentile=0.95, cond_drop_prob=0.1):
        super().__init__()
        self.condition_on_text = condition_on_text
        self.unconditional = not condition_on_text
        self.channels = channels
        unets = cast_tuple(unets)
        n = len(unets)
        timesteps = cast_tuple(timesteps, n)
        ns = cast_tuple(noise_schedules)
        ns = (*ns, *("cosine",) * max(0, 2 - len(ns)))
        ns = (*ns, *("linear",) * max(0, n - len(ns)))
        self.noise_schedulers = [GaussianDiffusionContinuousTimes(noise_schedule=s, timesteps=t)
                                 for t, s in zip(timesteps, ns)]
        self.lowres_noise_schedule = GaussianDiffusionContinuousTimes(noise_schedule=lowres_noise_schedule)
        self.pred_objectives = cast_tuple(pred_objectives, n)
        self.text_embed_dim = default(text_embed_dim, 768)
        self.unets = nn.ModuleList([])
        for ind, u in enumerate(unets):
            assert isinstance(u, (Unet, NullUnet))
            u = u.cast_model_parameters(
                lowres_cond=ind != 0, cond_on_text=condition_on_text,
                text_embed_dim=self.text_embed_dim if condition_on_text else None,
                channels=channels, channels_out=channels)
            self.unets.append(u)
        self.image_sizes = cast_tuple(image_sizes)
        assert len(self.image_sizes) == n
        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.dynamic_thresholding = cast_tuple(dynamic_thresholding, n)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        self.cond_drop_prob = cond_drop_prob

    @staticmethod
    def normalize_img(img):
        return img * 2 - 1

    @staticmethod
    def unnormalize_img(img):
        return (img + 1) * 0.5

    def p_mean_variance(self, unet, x, t, *, noise_scheduler, t_next, text_embeds, text_mask,
                        cond_images, lowres_cond_img, lowres_noise_times, cond_scale,
                        pred_objective, dynamic_threshold):
        pred = unet.forward_with_cond_scale(
            x, noise_scheduler.log_snr(t), text_embeds=text_embeds, text_mask=text_mask,
            cond_images=cond_images, cond_scale=cond_scale, lowres_cond_img=lowres_cond_img,
            lowres_noise_times=(self.lowres_noise_schedule.log_snr(lowres_noise_times)
                                if exists(lowres_noise_times) else None))
        if pred_objective == "noise":
            x_start = noise_scheduler.predict_start_from_noise(x, t, pred)
        elif pred_objective == "v":
            x_start = noise_scheduler.predict_start_from_v(x, t, pred)
        elif pred_objective == "x_start":
            x_start = pred
        else:
            raise ValueError(pred_objective)
        if dynamic_threshold:
            s = torch.quantile(x_start.flatten(1).abs(), self.dynamic_thresholding_percentile, dim=-1)
            s = _pad(x_start, s.clamp(min=1.0))
            x_start = x_start.clamp(-s, s) / s
        else:
            x_start = x_start.clamp(-1.0, 1.0)
        return noise_scheduler.q_posterior(x_start=x_start, x_t=x, t=t, t_next=t_next), x_start

    def p_sample(self, unet, x, t, noise, **kw):
        (mean, _, log_var), x_start = self.p_mean_variance(unet, x, t, **kw)
        b = x.shape[0]
        nonzero = (1 - (kw["t_next"] == 0).float()).reshape(b, *((1,) * (x.ndim - 1)))
        return mean + nonzero * (0.5 * log_var).exp() * noise, x_start

    def p_sample_loop(self, unet, shape, *, stage, noise_fn, noise_scheduler, lowres_cond_img,
                      lowres_noise_times, text_embeds, text_mask, cond_images, inpaint_images,
                      inpaint_masks, inpaint_resample_times, cond_scale, pred_objective,
                      dynamic_threshold, trace=None):
        batch = shape[0]
        img = noise_fn(("init", stage), shape)
        has_inpainting = exists(inpaint_images) and exists(inpaint_masks)
        resample_times = inpaint_resample_times if has_inpainting else 1
        if has_inpainting:
            inpaint_images = self.normalize_img(inpaint_images)
            inpaint_images = resize_image_to(inpaint_images, shape[-1])
            inpaint_masks = resize_image_to(inpaint_masks[:, None].float(), shape[-1]).bool()
        kw = dict(noise_scheduler=noise_scheduler, text_embeds=text_embeds, text_mask=text_mask,
                  cond_images=cond_images, lowres_cond_img=lowres_cond_img,
                  lowres_noise_times=lowres_noise_times, cond_scale=cond_scale,
                  pred_objective=pred_objective, dynamic_threshold=dynamic_threshold)
        for k, (times, times_next) in enumerate(noise_scheduler.get_sampling_timesteps(batch)):
            is_last_timestep = times_next == 0
            for r in reversed(range(resample_times)):
                if has_inpainting:
                    noised = noise_scheduler.q_sample(inpaint_images, times, noise_fn(("inpaint", stage, k, r), shape))
                    img = img * ~inpaint_masks + noised * inpaint_masks
                img, x_start = self.p_sample(unet, img, times, noise_fn(("step", stage, k, r), shape),
                                             t_next=times_next, **kw)
                if has_inpainting and not (r == 0 or bool(torch.all(is_last_timestep))):
                    renoised = noise_scheduler.q_sample_from_to(
                        img, times_next, times, noise_fn(("renoise", stage, k, r), shape))
                    img = torch.where(_pad(img, is_last_timestep), img, renoised)
            if exists(trace):  # state after timestep k (all resamples done)
                trace.append(img.clone())
        img = img.clamp(-1.0, 1.0)
        if has_inpainting:
            img = img * ~inpaint_masks + inpaint_images * inpaint_masks
        return self.unnormalize_img(img)

    @torch.no_grad()
    def sample(self, *, noise_fn, text_embeds=None, text_masks=None, cond_images=None,
               inpaint_images=None, inpaint_masks=None, inpaint_resample_times=5, batch_size=1,
               cond_scale=1.0, lowres_sample_noise_level=None, start_at_unet_number=1,
               start_image_or_video=None, stop_at_unet_number=None, return_all_unet_outputs=False,
               trace=None):
        self.eval()
        if not self.unconditional:
            assert exists(text_embeds)
            text_masks = default(text_masks, lambda: torch.any(text_embeds != 0.0, dim=-1))
            batch_size = text_embeds.shape[0]
        if exists(inpaint_images):
            if self.unconditional and batch_size == 1:
                batch_size = inpaint_images.shape[0]
            assert inpaint_images.shape[0] == batch_size
        assert not (self.condition_on_text and not exists(text_embeds))
        assert not (not self.condition_on_text and exists(text_embeds))
        assert not (exists(inpaint_images) ^ exists(inpaint_masks))
        lowres_sample_noise_level = default(lowres_sample_noise_level, self.lowres_sample_noise_level)
        n = len(self.unets)
        cond_scale = cast_tuple(cond_scale, n)
        outputs, img = [], None
        if start_at_unet_number > 1:
            assert exists(start_image_or_video)
            img = resize_image_to(start_image_or_video, self.image_sizes[start_at_unet_number - 2])
        for num, unet, size, sched, obj, dyn, cs in zip(
                range(1, n + 1), self.unets, self.image_sizes, self.noise_schedulers,
                self.pred_objectives, self.dynamic_thresholding, cond_scale):
            if num < start_at_unet_number:
                continue
            assert not isinstance(unet, NullUnet), "cannot sample from a placeholder unet"
            lowres_cond_img = lowres_noise_times = None
            shape = (batch_size, self.channels, size, size)
            if unet.lowres_cond:
                lowres_noise_times = self.lowres_noise_schedule.get_times(batch_size, lowres_sample_noise_level)
                lowres_cond_img = self.normalize_img(resize_image_to(img, size))
                lowres_cond_img = self.lowres_noise_schedule.q_sample(
                    lowres_cond_img, lowres_noise_times, noise_fn(("lowres", num), lowres_cond_img.shape))
            img = self.p_sample_loop(
                unet, shape, stage=num, noise_fn=noise_fn, noise_scheduler=sched,
                lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times,
                text_embeds=text_embeds, text_mask=text_masks, cond_images=cond_images,
                inpaint_images=inpaint_images, inpaint_masks=inpaint_masks,
                inpaint_resample_times=inpaint_resample_times, cond_scale=cs,
                pred_objective=obj, dynamic_threshold=dyn, trace=trace)
            outputs.append(img)
            if exists(stop_at_unet_number) and stop_at_unet_number == num:
                break
        return outputs if return_all_unet_outputs else outputs[-1]
