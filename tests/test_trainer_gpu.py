"""The `ImagenTrainer` entry of the sampling path on the GPU (sample_uncond.py:22-55, sample_cond.py:26-48:
construct the trainer around an Imagen, `trainer.load(path)`, `trainer.sample(...)` - which samples from the
EMA weights), plus the module-level pieces around it: deep copies of live UNets, stale-plan detection after
in-place weight changes, `Unet.forward_with_cond_scale`."""
import copy

import pytest
import torch

import helpers as H
from oracle import imagen_ref as R
from oracle import sampler_ref as RS

pytestmark = pytest.mark.gpu

FWD_REL_L2 = 2e-5
SAMPLE_ABS = 2e-3


def _pair(device, seed=5, T=4):
    import imagen_pytorch as ip

    ou = H.oracle_unet("small1", seed=seed)
    oim = RS.Imagen([ou], image_sizes=(16,), timesteps=(T,), pred_objectives=("noise",), condition_on_text=False)
    pim = ip.Imagen([ip.Unet(**oim.unets[0]._locals)], image_sizes=(16,), timesteps=(T,), pred_objectives=("noise",),
                    condition_on_text=False)
    pim.load_state_dict(oim.state_dict(), strict=True)
    return oim, pim.to(device)


def test_trainer_load_and_sample_use_the_ema_weights(device, tmp_path):
    """Checkpoint {'model','ema','version','steps'} in the trainer's layout -> trainer.load -> trainer.sample must
    equal the oracle sampling with the EMA weights (and differ from the online weights)."""
    import imagen_pytorch as ip
    from imagen_pytorch.version import __version__

    oim_online, pim = _pair(device, seed=5)
    oim_ema, _ = _pair(device, seed=6)          # a second set of weights plays the EMA copy
    ema = {f"0.ema_model.{k[len('unets.0.'):]}": v for k, v in oim_ema.state_dict().items() if k.startswith("unets.0.")}
    ema.update({f"0.online_model.{k[len('unets.0.'):]}": v for k, v in oim_online.state_dict().items()
                if k.startswith("unets.0.")})
    ema.update({"0.initted": torch.tensor([True]), "0.step": torch.tensor([7])})
    path = tmp_path / "unet1.pt"
    torch.save({"model": oim_online.state_dict(), "ema": ema, "version": __version__, "steps": torch.tensor([7])}, path)

    # the reference builds the trainer around a fresh Imagen (sample_uncond.py:22-23); a LIVE one - already sampled,
    # so its UNet holds engine handles - must work too (copy.deepcopy of the UNets for the EMA copies)
    pim.sample(batch_size=1, seed=1, device=device)
    assert pim.unets[0]._engines
    trainer = ip.ImagenTrainer(imagen=pim)
    assert not trainer.ema_unets[0]._engines and pim.unets[0]._engines
    trainer.load(str(path))
    assert int(trainer.steps[0]) == 7
    nf = RS.generator_noise_fn(11)
    ref_ema = oim_ema.sample(noise_fn=nf, batch_size=2)
    ref_online = oim_online.sample(noise_fn=nf, batch_size=2)
    got = trainer.sample(batch_size=2, noise_fn=nf)          # device defaults to the trainer's
    assert got.is_cuda and (got.cpu() - ref_ema).abs().max() < SAMPLE_ABS
    assert (ref_ema - ref_online).abs().max() > 10 * SAMPLE_ABS
    # outside the trainer the Imagen still samples from the online weights
    assert trainer.imagen.unets is not trainer.ema_unets
    got_online = pim.sample(batch_size=2, noise_fn=nf, device=device)
    assert (got_online.cpu() - ref_online).abs().max() < SAMPLE_ABS
    # a checkpoint without an 'ema' section: the trainer samples from the loaded online weights
    torch.save({"model": oim_ema.state_dict(), "version": __version__}, path)
    trainer.load(str(path))
    got2 = trainer.sample(batch_size=2, noise_fn=nf)
    assert (got2.cpu() - ref_ema).abs().max() < SAMPLE_ABS


def test_trainer_sample_in_chunks_of_max_batch_size(device):
    """`trainer.sample(max_batch_size=)` - the library's imagen_sample_in_chunks: the batch is cut into chunks of at most
    that size, batched tensor arguments alongside, outputs joined in order."""
    import imagen_pytorch as ip

    oim, pim = _pair(device, seed=8, T=3)
    trainer = ip.ImagenTrainer(imagen=pim, use_ema=False)
    nf = RS.generator_noise_fn(21)
    want = torch.cat([oim.sample(noise_fn=nf, batch_size=b) for b in (2, 2, 1)])
    got = trainer.sample(batch_size=5, max_batch_size=2, noise_fn=nf)
    assert got.shape == want.shape and (got.cpu() - want).abs().max() < SAMPLE_ABS
    # batched tensor arguments are cut with the batch: inpainting a known half
    inp = torch.rand(3, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    mask = torch.zeros(3, 16, 16)
    mask[:, :, :8] = 1
    want = torch.cat([oim.sample(noise_fn=nf, batch_size=b, inpaint_images=inp[a:a + b], inpaint_masks=mask[a:a + b],
                                 inpaint_resample_times=2) for a, b in ((0, 2), (2, 1))])
    got = trainer.sample(batch_size=3, max_batch_size=2, noise_fn=nf, inpaint_images=inp.to(device),
                         inpaint_masks=mask.to(device), inpaint_resample_times=2)
    assert (got.cpu() - want).abs().max() < SAMPLE_ABS
    assert torch.equal(got.cpu()[:, :, :, :8], inp[:, :, :, :8])   # known pixels are pasted back exactly
    # a chunk size that covers the batch is a plain call
    one = trainer.sample(batch_size=2, max_batch_size=4, noise_fn=nf)
    assert (one.cpu() - oim.sample(noise_fn=nf, batch_size=2)).abs().max() < SAMPLE_ABS


def test_deepcopy_and_pickle_of_a_live_unet_drop_the_engine(device):
    _, pim = _pair(device)
    u = pim.unets[0]
    x, t = torch.randn(1, 3, 16, 16, device=device), torch.zeros(1, device=device)
    y = u(x, t)
    c = copy.deepcopy(u)
    assert u._engines and not c._engines and not c._io_buffers
    assert torch.equal(c(x, t), y)
    import io

    buf = io.BytesIO()
    torch.save(u, buf)       # whole-module pickle (the library's checkpoint-path construction does this)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert not r._engines and torch.equal(r.to(device)(x, t), y)
    del c, r                 # each owner destroys only its own plans
    assert torch.equal(u(x, t), y)


def test_in_place_weight_changes_rebuild_the_packed_copies(device):
    """The engine holds packed COPIES of the weights.  restore_parts() on a live state_dict, p.copy_() under
    no_grad or an optimizer step change the parameters in place without load_state_dict: the next forward must
    see the new values (and match the oracle on them)."""
    from imagen_pytorch import restore_parts

    oim_a, pim = _pair(device, seed=5)
    oim_b, _ = _pair(device, seed=9)
    ua, ub, pu = oim_a.unets[0], oim_b.unets[0], pim.unets[0]
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(2, 3, 16, 16, generator=g), torch.randn(2, generator=g)
    with torch.no_grad():
        ref_a, ref_b = ua(x, t), ub(x, t)
    assert H.rel_l2(pu(x.to(device), t.to(device)), ref_a) < FWD_REL_L2
    restore_parts(pim.state_dict(), oim_b.state_dict())      # in place, no load_state_dict afterwards
    assert H.rel_l2(pu(x.to(device), t.to(device)), ref_b) < FWD_REL_L2
    with torch.no_grad():                                     # a single tensor written in place
        pu.final_conv.bias.copy_(ua.final_conv.bias)
        ub.final_conv.bias.copy_(ua.final_conv.bias)
        ref_c = ub(x, t)
    got = pu(x.to(device), t.to(device))
    assert H.rel_l2(got, ref_c) < FWD_REL_L2 and H.rel_l2(got, ref_b) > 1e-4
    # writes through .data have their own version counter: the documented contract is invalidate_engine()
    with torch.no_grad():
        pu.final_conv.bias.data.copy_(ub.final_conv.bias.data * 0 + 0.25)
        ub.final_conv.bias.fill_(0.25)
        ref_d = ub(x, t)
    pu.invalidate_engine()
    got = pu(x.to(device), t.to(device))
    assert H.rel_l2(got, ref_d) < FWD_REL_L2
    # an unchanged model keeps its plan (no rebuild per call)
    h0 = list(pu._engines.values())[0].value
    pu(x.to(device), t.to(device))
    assert list(pu._engines.values())[0].value == h0


def test_unet_forward_with_cond_scale_matches_oracle(device):
    """Unet.forward_with_cond_scale(cond_scale != 1): two forwards (conditioning kept / dropped) and
    null + (cond - null) * scale - the module-level method behind sample.py:55-59."""
    import imagen_pytorch as ip

    kw = dict(dim=32, dim_mults=(1, 2, 3, 4), cond_dim=64, text_embed_dim=3, num_resnet_blocks=2,
              layer_attns=(False, True, True, True), layer_cross_attns=(False, True, True, True),
              cond_images_channels=4)   # train.py:30-39 at reduced dim
    ou = H.randomize_(R.Unet(**kw, cond_on_text=True), 23).eval()
    pu = ip.Unet(**ou._locals)
    pu.load_state_dict(ou.state_dict(), strict=True)
    pu = pu.to(device)
    B, S = 2, 16
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randn(B, generator=g)
    text = torch.tensor([0.0, 0.5, 0.2]).reshape(1, 1, 3).repeat_interleave(B, dim=0)
    mask = torch.any(text != 0.0, dim=-1)
    labels = torch.nn.functional.one_hot(torch.randint(0, 4, (B, S, S), generator=g), 4).permute(0, 3, 1, 2).float()
    dv = lambda v: v.to(device)
    for cs in (1.0, 2.5):
        with torch.no_grad():
            ref = ou.forward_with_cond_scale(x, t, text_embeds=text, text_mask=mask, cond_images=labels, cond_scale=cs)
        got = pu.forward_with_cond_scale(dv(x), dv(t), text_embeds=dv(text), text_mask=dv(mask),
                                         cond_images=dv(labels), cond_scale=cs)
        assert H.rel_l2(got, ref) < FWD_REL_L2, (cs, H.rel_l2(got, ref))
