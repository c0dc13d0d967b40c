"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

IMPORTANT: these vectors pin the BUILD'S OWN CPU restatement (oracle/), not the third-party library
the reference calls (imagen-pytorch 1.18.5 is absent from this image and the reference ships no
tests, seeds, checkpoints or golden outputs — SURVEY.md §4, §8c: PARITY UNPINNED).  They freeze the
oracle so that a later edit of oracle/ or of the engine cannot drift silently.

    python tests/golden/make_golden.py          # rewrites the .npz files
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import helpers as H  # noqa: E402
from oracle import sampler_ref as RS  # noqa: E402

OUT = Path(__file__).resolve().parent


def unet_forward_case(name, lowres, B, S, seed):
    ou = H.oracle_unet(name, lowres_cond=lowres, seed=seed).eval()
    g = torch.Generator().manual_seed(seed + 100)
    kw = H.UNET_KW[name]
    x = torch.randn(B, 3, S, S, generator=g)
    lr = torch.randn(B, 3, S, S, generator=g) if lowres else None
    cc = kw.get("cond_images_channels", 0)
    cond = torch.rand(B, cc, S, S, generator=g) if cc else None
    t = torch.randn(B, generator=g) * 3
    tl = torch.full((B,), -1.3) if lowres else None
    with torch.no_grad():
        y = ou(x, t, lowres_cond_img=lr, lowres_noise_times=tl, cond_images=cond)
    wsum = float(sum(p.detach().double().abs().sum() for p in ou.parameters()))
    d = dict(x=x.numpy(), t=t.numpy(), y=y.numpy(), weight_abs_sum=np.float64(wsum), seed=np.int64(seed))
    if lowres:
        d.update(lowres=lr.numpy(), t_lowres=tl.numpy())
    if cc:
        d.update(cond=cond.numpy())
    return d


def sampler_case(seed):
    ous = [H.oracle_unet("small1", seed=seed), H.oracle_unet("small2", lowres_cond=True, seed=seed + 1)]
    oim = RS.Imagen(ous, image_sizes=(16, 32), timesteps=(5, 4), pred_objectives=("noise", "v"),
                    condition_on_text=False)
    g = torch.Generator().manual_seed(seed + 7)
    cond = torch.rand(1, 3, 32, 32, generator=g)
    inp = torch.rand(1, 3, 32, 32, generator=g)
    mask = torch.zeros(1, 32, 32)
    mask[:, :8] = 1
    nf = RS.generator_noise_fn(seed)
    base = oim.sample(noise_fn=nf, batch_size=1, stop_at_unet_number=1)
    sr = oim.sample(noise_fn=nf, batch_size=1, cond_images=cond, start_image_or_video=base, start_at_unet_number=2,
                    inpaint_images=inp, inpaint_masks=mask, inpaint_resample_times=2)
    return dict(cond=cond.numpy(), inpaint=inp.numpy(), mask=mask.numpy(), base=base.numpy(), sr=sr.numpy(),
                seed=np.int64(seed))


def schedule_case():
    out = {}
    for name, T in (("cosine", 250), ("linear", 256)):
        s = RS.GaussianDiffusionContinuousTimes(noise_schedule=name, timesteps=T)
        ts = torch.linspace(1.0, 0.0, T + 1)
        out[f"{name}_log_snr"] = s.log_snr(ts).numpy()
    return out


def main():
    torch.set_num_threads(4)
    np.savez_compressed(OUT / "unet_small1.npz", **unet_forward_case("small1", False, 2, 16, 21))
    np.savez_compressed(OUT / "unet_small2_lowres.npz", **unet_forward_case("small2", True, 1, 32, 22))
    np.savez_compressed(OUT / "sampler_cascade.npz", **sampler_case(31))
    np.savez_compressed(OUT / "schedules.npz", **schedule_case())
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
