"""Sampling-side surface of ``imagen_pytorch.trainer`` as the reference uses it:

  * ``restore_parts``      sample_ultra_res.py:12,63 / outpainting.py (partial state-dict load)
  * ``ImagenTrainer(imagen=...)`` + ``.load(path)`` + ``.sample(**kwargs)``
                           sample_uncond.py:22-55, sample_cond.py:26-48, sample.py:23-60

Training methods are outside the hot path and raise ``NotImplementedError``.
"""
from __future__ import annotations

import copy
from contextlib import contextmanager
from pathlib import Path

import torch
from packaging import version
from torch import nn

from .imagen_pytorch import Imagen, NullUnet, exists
from .version import __version__


def restore_parts(state_dict_target, state_dict_from, report=print):
    """Copies every same-named, same-shaped tensor and returns the target (the reference falls
    back to it when the strict load raises, sample_ultra_res.py:59-63).  The library only prints shape
    mismatches; a checkpoint written by another imagen-pytorch version can also carry keys this module tree
    does not have (e.g. `q_scale` / `k_scale` of the qk-norm attention) or lack keys it has - both would
    leave the model silently half-loaded, so they are reported too (`report=None` silences it)."""
    say = report if report is not None else (lambda *_: None)
    unexpected, copied = [], 0
    for name, param in state_dict_from.items():
        if name not in state_dict_target:
            unexpected.append(name)
            continue
        if param.size() == state_dict_target[name].size():
            state_dict_target[name].copy_(param)
            copied += 1
        else:
            say(f"layer {name}({param.size()} different than target: {state_dict_target[name].size()}")
    missing = [k for k in state_dict_target if k not in state_dict_from]
    if unexpected or missing:
        say(f"restore_parts: {copied} tensors copied; {len(unexpected)} checkpoint tensors have no counterpart in "
            f"this model (ignored): {_abbrev(unexpected)}; {len(missing)} model tensors are not in the checkpoint "
            f"(left at their initial values): {_abbrev(missing)}")
        if any(k.endswith(("q_scale", "k_scale")) for k in unexpected):
            say("restore_parts: the checkpoint has q_scale / k_scale tensors: it was written by an imagen-pytorch "
                "whose attention normalises q and k (qk-norm); load it with load_state_dict, which switches the "
                "attention variant from the key set (Unet(attn_qk_norm=...))")
    return state_dict_target


def _abbrev(keys, n=6):
    return "[]" if not keys else "[" + ", ".join(keys[:n]) + (f", ... +{len(keys) - n} more]" if len(keys) > n else "]")


def _open(path):
    try:
        from fsspec.core import url_to_fs

        fs, _ = url_to_fs(str(path))
        return fs, fs.open(str(path))
    except ImportError:  # plain local files still work
        return None, open(path, "rb")


class ImagenTrainer(nn.Module):
    def __init__(self, imagen=None, imagen_checkpoint_path=None, use_ema=True, lr=1e-4, fp16=False,
                 max_grad_norm=None, dl_tuple_output_keywords_names=("images", "text_embeds", "text_masks",
                                                                      "cond_images"),
                 **ignored_training_kwargs):
        super().__init__()
        assert exists(imagen) ^ exists(imagen_checkpoint_path), \
            "either imagen instance is passed into the trainer, or a checkpoint path that contains the imagen config"
        assert isinstance(imagen, Imagen), "checkpoint-path construction is not used by the reference"
        if fp16:
            raise NotImplementedError("the engine computes in fp32, as the reference samples (fp16=False)")
        self.imagen = imagen
        self.use_ema = use_ema
        self.num_unets = len(imagen.unets)
        # EMA copies, filled by load(); sampling through the trainer uses them
        self.ema_unets = nn.ModuleList([copy.deepcopy(u) for u in imagen.unets]) if use_ema else None
        self.register_buffer("steps", torch.tensor([0] * self.num_unets))
        self.register_buffer("_temp", torch.tensor([0.0]), persistent=False)

    @property
    def device(self):
        return self.imagen.device

    # ---- checkpoint
    def load(self, path, only_model=False, strict=True, noop_if_not_exist=False):
        fs, f = _open(path)
        if noop_if_not_exist and ((fs is not None and not fs.exists(str(path))) or
                                  (fs is None and not Path(path).exists())):
            print(f"trainer checkpoint not found at {path}")
            return
        with f:
            loaded_obj = torch.load(f, map_location="cpu")
        if "version" in loaded_obj and version.parse(__version__) != version.parse(loaded_obj["version"]):
            print(f'loading saved imagen at version {loaded_obj["version"]}, but current package version is {__version__}')
        try:
            self.imagen.load_state_dict(loaded_obj["model"], strict=strict)
        except RuntimeError:
            print("Failed loading state dict. Trying partial load")
            self.imagen.load_state_dict(restore_parts(self.imagen.state_dict(), loaded_obj["model"]))
        if only_model:
            return loaded_obj
        if "steps" in loaded_obj:
            self.steps.copy_(loaded_obj["steps"])
        if self.use_ema:
            ema_sd = loaded_obj.get("ema")
            if exists(ema_sd):
                self._load_ema(ema_sd)
            else:  # no EMA section: sample from the online weights
                for e, u in zip(self.ema_unets, self.imagen.unets):
                    e.load_state_dict(u.state_dict())
        return loaded_obj

    def _load_ema(self, ema_sd):
        """'ema' is the state_dict of a ModuleList of ema-pytorch wrappers: keys
        ``{i}.ema_model.<unet key>`` (+ ``{i}.online_model.*``, ``{i}.initted``, ``{i}.step``)."""
        for i, e in enumerate(self.ema_unets):
            prefix = f"{i}.ema_model."
            sub = {k[len(prefix):]: v for k, v in ema_sd.items() if k.startswith(prefix)}
            if not sub or isinstance(e, NullUnet):
                continue
            try:
                e.load_state_dict(sub, strict=True)
            except RuntimeError:
                print(f"Failed loading EMA state dict of unet {i + 1}. Trying partial load")
                e.load_state_dict(restore_parts(e.state_dict(), sub))

    def save(self, *a, **k):
        raise NotImplementedError("training-side checkpoint writing is outside the sampling hot path")

    # ---- sampling
    @contextmanager
    def use_ema_unets(self):
        if not self.use_ema:
            yield
            return
        online = self.imagen.unets
        self.imagen.unets = self.ema_unets
        try:
            yield
        finally:
            self.imagen.unets = online

    @torch.no_grad()
    def sample(self, *args, **kwargs):
        """`ImagenTrainer.sample` (sample_uncond.py:49-55, sample_cond.py:40-48): EMA weights, the trainer's device.
        `max_batch_size` splits the call the way the library's `imagen_sample_in_chunks` does: `batch_size` (or the
        leading dimension of the batched tensor arguments) is cut into chunks of at most `max_batch_size`, every
        tensor argument whose leading dimension equals the batch is cut alongside, and the outputs are joined in
        order (tensors concatenated, PIL lists chained, per-unet lists joined per unet)."""
        kwargs.setdefault("device", self.device)
        max_batch_size = kwargs.pop("max_batch_size", None)
        with self.use_ema_unets():
            if not exists(max_batch_size):
                return self.imagen.sample(*args, **kwargs)
            return self._sample_in_chunks(int(max_batch_size), *args, **kwargs)

    def _sample_in_chunks(self, max_batch_size, *args, **kwargs):
        assert max_batch_size >= 1
        if self.imagen.unconditional:
            batch_size = kwargs.get("batch_size")
            if batch_size is None:   # positional form of the library: sample(texts, text_masks, ..., batch_size at 14)
                batched = [a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a) and a.dim() > 0]
                batch_size = batched[0].shape[0] if batched else 1
        else:
            batched = [a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a) and a.dim() > 0]
            assert batched, "a text-conditioned sample() needs its text embeddings"
            batch_size = batched[0].shape[0]
        sizes = [max_batch_size] * (batch_size // max_batch_size)
        if batch_size % max_batch_size:
            sizes.append(batch_size % max_batch_size)
        if len(sizes) <= 1:
            return self.imagen.sample(*args, **kwargs)

        def cut(v, n0, n1):
            return v[n0:n1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == batch_size else v

        seed = kwargs.get("seed")
        outs, n0 = [], 0
        for k, b in enumerate(sizes):
            kw = {name: cut(v, n0, n0 + b) for name, v in kwargs.items()}
            if "batch_size" in kw or self.imagen.unconditional:
                kw["batch_size"] = b
            if seed is not None:   # a seeded call stays reproducible and its chunks draw different noise
                kw["seed"] = int(seed) + 1000003 * k
            outs.append(self.imagen.sample(*[cut(a, n0, n0 + b) for a in args], **kw))
            n0 += b
        first = outs[0]
        if torch.is_tensor(first):
            return torch.cat(outs, dim=0)
        if isinstance(first, (list, tuple)) and first and torch.is_tensor(first[0]):   # return_all_unet_outputs
            return [torch.cat([o[i] for o in outs], dim=0) for i in range(len(first))]
        if isinstance(first, (list, tuple)) and first and isinstance(first[0], (list, tuple)):   # ... as PIL lists per unet
            return [[im for o in outs for im in o[i]] for i in range(len(first))]
        return [im for o in outs for im in o]   # PIL images

    # ---- training surface (out of scope)
    def _no_training(self, *a, **k):
        raise NotImplementedError("training is outside the sampling hot path this package replaces (SURVEY §2)")

    train_step = valid_step = add_train_dataset = add_valid_dataset = add_train_dataloader = _no_training
    add_valid_dataloader = update = forward = _no_training
