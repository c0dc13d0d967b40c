"""CPU tier (`-m "not gpu"`): the oracle against the committed golden vectors, the host-side logic
of the drop-in package, and the C ABI (library loads, exports every declared symbol).  No compute
call touches the engine here."""
import ctypes as C
import io
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers as H
from oracle import imagen_ref as R
from oracle import sampler_ref as RS

ROOT = Path(__file__).resolve().parent.parent
GOLD = Path(__file__).resolve().parent / "golden"


def _t(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------------------- oracle vs golden
@pytest.mark.parametrize("fname,name,lowres", [("unet_small1.npz", "small1", False),
                                               ("unet_small2_lowres.npz", "small2", True)])
def test_oracle_unet_forward_matches_golden(fname, name, lowres):
    g = np.load(GOLD / fname)
    ou = H.oracle_unet(name, lowres_cond=lowres, seed=int(g["seed"])).eval()
    wsum = float(sum(p.detach().double().abs().sum() for p in ou.parameters()))
    assert abs(wsum - float(g["weight_abs_sum"])) < 1e-6 * wsum, "seeded weight generation changed"
    with torch.no_grad():
        y = ou(_t(g["x"]), _t(g["t"]), lowres_cond_img=_t(g["lowres"]) if lowres else None,
               lowres_noise_times=_t(g["t_lowres"]) if lowres else None,
               cond_images=_t(g["cond"]) if "cond" in g else None)
    assert H.rel_l2(y, _t(g["y"])) < 1e-5  # thread-count dependent summation order only


def test_oracle_cascade_sampler_matches_golden():
    g = np.load(GOLD / "sampler_cascade.npz")
    seed = int(g["seed"])
    ous = [H.oracle_unet("small1", seed=seed), H.oracle_unet("small2", lowres_cond=True, seed=seed + 1)]
    oim = RS.Imagen(ous, image_sizes=(16, 32), timesteps=(5, 4), pred_objectives=("noise", "v"),
                    condition_on_text=False)
    nf = RS.generator_noise_fn(seed)
    base = oim.sample(noise_fn=nf, batch_size=1, stop_at_unet_number=1)
    assert (base - _t(g["base"])).abs().max() < 1e-4
    sr = oim.sample(noise_fn=nf, batch_size=1, cond_images=_t(g["cond"]), start_image_or_video=_t(g["base"]),
                    start_at_unet_number=2, inpaint_images=_t(g["inpaint"]), inpaint_masks=_t(g["mask"]),
                    inpaint_resample_times=2)
    assert (sr - _t(g["sr"])).abs().max() < 1e-4
    m = _t(g["mask"]).bool()[:, None].expand_as(sr)
    assert torch.equal(sr[m], _t(g["inpaint"])[m])  # known pixels pasted back exactly


def test_schedules_match_golden_and_product_tables_match_oracle_scalars():
    from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes as PS

    g = np.load(GOLD / "schedules.npz")
    for name, T in (("cosine", 250), ("linear", 256)):
        o = RS.GaussianDiffusionContinuousTimes(noise_schedule=name, timesteps=T)
        ts = torch.linspace(1.0, 0.0, T + 1)
        assert np.allclose(o.log_snr(ts).numpy(), g[f"{name}_log_snr"], rtol=1e-6, atol=1e-6)
        tb = PS(noise_schedule=name, timesteps=T).step_tables()
        x = torch.ones(1, 1)
        for k, (t, tn) in enumerate(o.get_sampling_timesteps(1)):
            mean, var, logvar = o.q_posterior(x_start=x * 0.25, x_t=x, t=t, t_next=tn)
            a, an, c = tb["alpha"][k], tb["alpha_next"][k], tb["c"][k]
            mine = an * (1.0 * (1 - c) / a + c * 0.25)
            assert abs(float(mean) - float(mine)) <= 1e-6 * abs(float(mean)) + 1e-9
            ns = (1 - float(tn == 0)) * float((0.5 * logvar).exp())
            assert abs(ns - float(tb["noise_scale"][k])) <= 1e-6 * abs(ns) + 1e-12
            # re-noise t_next -> t
            rn = o.q_sample_from_to(x, tn, t, noise=x * 2)
            if k < T - 1:
                assert abs(float(rn) - float(tb["rn_a"][k] + 2 * tb["rn_b"][k])) < 1e-5
        assert tb["noise_scale"][-1] == 0 and (tb["noise_scale"][:-1] > 0).all()


def test_quantile_dynamic_threshold_semantics_of_oracle():
    """s = max(1, q95|x0|) per sample; x0 clamped to [-s,s] then divided by s."""
    oim = RS.Imagen([R.NullUnet()], image_sizes=(8,), timesteps=2, condition_on_text=False)
    x0 = torch.linspace(-4, 4, 2 * 3 * 8 * 8).reshape(2, 3, 8, 8)
    s = torch.quantile(x0.flatten(1).abs(), 0.95, dim=-1).clamp(min=1.0)
    out = x0.clamp(-s[:, None, None, None], s[:, None, None, None]) / s[:, None, None, None]
    assert out.abs().max() <= 1.0 + 1e-6 and oim.dynamic_thresholding_percentile == 0.95


# ------------------------------------------------------------------------------- host logic (product package)
REFERENCE_CONFIGS = {
    # name -> (Unet kwargs at full size, lowres)     [reference file:line]
    "ultra_unet1": (dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                         layer_cross_attns=(False, True, True, True), cond_images_channels=3), False),  # train_ultra_res.py:29-36
    "ultra_unet2": (dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                         layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                         init_conv_to_final_conv_residual=True, cond_images_channels=3), True),  # :39-48
    "ultra_unet3": (dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True,
                         layer_attns=False, layer_cross_attns=(False, False, False, True),
                         init_conv_to_final_conv_residual=True, cond_images_channels=3), True),  # :51-60
    "uncond_unet1": (dict(dim=256, dim_mults=(1, 2, 4, 8), cond_dim=512, num_resnet_blocks=3,
                          layer_attns=(False, True, True, True), layer_cross_attns=(False, True, True, True)),
                     False),  # train_uncond.py:30-36
    "segcond_unet1": (dict(dim=256, dim_mults=(1, 2, 3, 4), cond_dim=512, text_embed_dim=3, num_resnet_blocks=3,
                           layer_attns=(False, True, True, True), layer_cross_attns=(False, True, True, True),
                           cond_images_channels=4, cond_on_text=True), False),  # train.py:30-39
    "v2_unet2": (dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                      layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                      init_conv_to_final_conv_residual=True, cond_images_channels=6), True),  # train_ultra_res_v2.py
}


@pytest.mark.parametrize("name", sorted(REFERENCE_CONFIGS))
def test_product_state_dict_layout_equals_oracle_for_reference_configs(name):
    """Checkpoint-layout contract: same keys, same shapes, for every UNet the reference builds."""
    import imagen_pytorch as ip

    kw, lowres = REFERENCE_CONFIGS[name]
    kw = dict(kw)
    cond_on_text = kw.pop("cond_on_text", False)
    ted = kw.pop("text_embed_dim", None)
    with torch.device("meta"):
        pu = ip.Unet(**kw, lowres_cond=lowres, cond_on_text=cond_on_text, text_embed_dim=ted)
        ou = R.Unet(**kw, lowres_cond=lowres, cond_on_text=cond_on_text, text_embed_dim=ted)
    ps = {k: tuple(v.shape) for k, v in pu.state_dict().items()}
    os_ = {k: tuple(v.shape) for k, v in ou.state_dict().items()}
    assert ps == os_
    # spot-check the recalled key shapes of SURVEY A.5
    for k in ("init_conv.convs.2.weight", "to_time_hiddens.0.weights", "norm_cond.weight", "null_text_embed",
              "mid_block1.cross_attn.to_kv.weight", "mid_attn.layers.0.0.to_out.1.g", "mid_attn.layers.0.1.4.weight",
              "final_res_block.gca.net.2.bias", "final_conv.bias", "downs.0.1.block1.groupnorm.weight",
              "ups.0.3.net.0.weight" if kw.get("memory_efficient") or True else ""):
        assert k in ps, k
    if lowres:
        assert "to_lowres_time_tokens.0.weight" in ps
    if kw.get("memory_efficient"):
        assert "init_resnet_block.gca.to_k.weight" in ps and "downs.0.0.1.weight" in ps
    else:
        assert "downs.3.4.fns.0.weight" in ps and "downs.0.4.1.weight" in ps


def test_reference_style_construction_and_checkpoint_roundtrip(tmp_path):
    """The reference's own construction pattern (train_ultra_res.py:65-92) and loader
    (sample_ultra_res.py:36-65) against the drop-in package, at reduced dim."""
    from imagen_pytorch import Imagen, ImagenTrainer, NullUnet, Unet, restore_parts
    from imagen_pytorch.version import __version__
    from torch import nn

    class FixedNullUnet(NullUnet):  # verbatim pattern of train_ultra_res.py:65-75
        def __init__(self, lowres_cond=False, *args, **kwargs):
            super().__init__()
            self.lowres_cond = lowres_cond
            self.dummy_parameter = nn.Parameter(torch.tensor([0.]))

        def cast_model_parameters(self, *args, **kwargs):
            return self

        def forward(self, x, *args, **kwargs):
            return x

    def init_imagen(unet_number):
        gen = lambda: Unet(**H.UNET_KW["small2"])
        return Imagen(
            unets=(Unet(**H.UNET_KW["small1"]) if unet_number == 1 else FixedNullUnet(),
                   gen() if unet_number == 2 else FixedNullUnet(lowres_cond=True),
                   gen() if unet_number == 3 else FixedNullUnet(lowres_cond=True)),
            image_sizes=(64, 256, 1024), timesteps=(1024, 256, 256), pred_objectives=("noise", "noise", "noise"),
            random_crop_sizes=(None, None, 256), condition_on_text=False)

    imagen = init_imagen(2)
    keys = list(imagen.state_dict().keys())
    assert keys[0] == "unets.0.dummy_parameter" and "unets.2.dummy_parameter" in keys
    assert any(k.startswith("unets.1.to_lowres_time_hiddens") for k in keys)  # Imagen re-cast unet 2 with lowres_cond
    assert not any("text_to_cond" in k for k in keys) and any("attn_pool.latents" in k for k in keys)
    assert [type(s.log_snr).__name__ for s in imagen.noise_schedulers]  # cosine, cosine, linear
    from imagen_pytorch.imagen_pytorch import alpha_cosine_log_snr, beta_linear_log_snr
    assert [s.log_snr for s in imagen.noise_schedulers] == [alpha_cosine_log_snr, alpha_cosine_log_snr,
                                                            beta_linear_log_snr]
    # checkpoint layout {'model','version',...}: strict load, then the RuntimeError -> restore_parts fallback
    path = tmp_path / "unet2_mag1.pt"
    torch.save({"model": imagen.state_dict(), "version": __version__, "steps": torch.tensor([0, 5, 0])}, path)
    other = init_imagen(2)
    obj = torch.load(path, map_location="cpu")
    other.load_state_dict(obj["model"], strict=True)
    for (ka, va), (kb, vb) in zip(imagen.state_dict().items(), other.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    wrong = init_imagen(3)  # unet number mismatch: strict load must raise RuntimeError (sample_ultra_res.py:59-63)
    with pytest.raises(RuntimeError):
        wrong.load_state_dict(obj["model"], strict=True)
    wrong.load_state_dict(restore_parts(wrong.state_dict(), obj["model"]))
    # trainer.load: model + EMA section in the ema-pytorch layout
    ema = {f"1.ema_model.{k[len('unets.1.'):]}": v + 1 for k, v in imagen.state_dict().items()
           if k.startswith("unets.1.")}
    ema.update({"1.initted": torch.tensor([True]), "1.step": torch.tensor([3])})
    torch.save({"model": imagen.state_dict(), "version": "1.18.0", "ema": ema, "steps": torch.tensor([0, 5, 0])}, path)
    tr = ImagenTrainer(imagen=init_imagen(2))
    tr.load(str(path))
    k = "final_conv.bias"
    assert torch.equal(tr.ema_unets[1].state_dict()[k], imagen.unets[1].state_dict()[k] + 1)
    assert torch.equal(tr.imagen.unets[1].state_dict()[k], imagen.unets[1].state_dict()[k])
    with tr.use_ema_unets():
        assert tr.imagen.unets is tr.ema_unets
    with pytest.raises(NotImplementedError):
        tr.train_step(unet_number=1)


@pytest.mark.parametrize("name", ["small1", "small2", "ultra1"])
def test_checkpoints_of_the_neighbouring_library_forks_load_strictly(name):
    """SURVEY A.1 lists two STRUCTURAL forks between library versions besides the attention similarity: Downsample as
    pixel-unshuffle + 1x1 conv vs Conv2d(4, stride 2, pad 1), and mid_attn as a TransformerBlock vs a bare residual
    attention.  A checkpoint names its fork through its keys and shapes; the drop-in Unet follows it, so the strict
    load of sample_ultra_res.py:59 succeeds instead of falling into restore_parts half-loaded."""
    from imagen_pytorch import Unet

    kw = dict(H.UNET_KW[name], cond_on_text=False, text_embed_dim=None)
    for forms in (dict(downsample_form="conv4x4"), dict(mid_attn_form="residual_attention"),
                  dict(downsample_form="conv4x4", mid_attn_form="residual_attention")):
        old = H.randomize_(R.Unet(**kw, **forms), 5)
        sd = old.state_dict()
        if "downsample_form" in forms:
            ds = [k for k, v in sd.items() if re.fullmatch(r"downs\.\d+\.[04]\.weight", k)]
            assert ds and all(sd[k].shape[-2:] == (4, 4) for k in ds) and not any(re.fullmatch(r"downs\.\d+\.[04]\.1\.weight", k) for k in sd)
        if "mid_attn_form" in forms:
            assert "mid_attn.fn.fn.to_q.weight" in sd and not any(k.startswith("mid_attn.layers") for k in sd)
        # a default-built product Unet and a default-built oracle both follow the checkpoint
        for cls in (Unet, R.Unet):
            u = cls(**kw)
            u.load_state_dict(sd, strict=True)
            assert u.downsample_form == forms.get("downsample_form", "unshuffle")
            assert u.mid_attn_form == forms.get("mid_attn_form", "transformer")
            assert list(u.state_dict().keys()) == list(sd.keys())
            assert all(torch.equal(a, b) for a, b in zip(u.state_dict().values(), sd.values()))
            # ... and back again with a checkpoint of the default fork
            new = R.Unet(**kw).state_dict()
            u.load_state_dict(new, strict=True)
            assert (u.downsample_form, u.mid_attn_form) == ("unshuffle", "transformer")
            assert list(u.state_dict().keys()) == list(new.keys())
        # the oracle's function really differs between the forks (no silent aliasing of the two module trees)
        x, t = torch.randn(1, 3, 16, 16), torch.randn(1)
        cond = torch.rand(1, 3, 16, 16) if kw.get("cond_images_channels") else None
        with torch.no_grad():
            a = old.eval()(x, t, cond_images=cond)
        assert torch.isfinite(a).all()


def test_qk_norm_fallback_follows_the_constructor():
    """A checkpoint without q_scale / k_scale resets a Unet that had switched to learned qk-norm - back to what the
    CONSTRUCTOR asked for (cosine_sim_attn=True stays cosine-sim), and a Unet built explicitly with attn_qk_norm=2
    keeps it: torch reports the missing keys (an error under strict=True)."""
    from imagen_pytorch import Unet

    kw = dict(H.UNET_KW["small1"], cond_on_text=False, text_embed_dim=None)
    with_scales = R.Unet(**kw, attn_qk_norm=2).state_dict()
    without = R.Unet(**kw).state_dict()
    u = Unet(**kw, cosine_sim_attn=True)
    assert u.attn_qk_norm == 1
    u.load_state_dict(with_scales, strict=True)
    assert u.attn_qk_norm == 2
    u.load_state_dict(without, strict=True)
    assert u.attn_qk_norm == 1, "cosine_sim_attn=True must survive a checkpoint without q_scale"
    u0 = Unet(**kw)
    u0.load_state_dict(with_scales, strict=True)
    u0.load_state_dict(without, strict=True)
    assert u0.attn_qk_norm == 0
    u2 = Unet(**kw, attn_qk_norm=2)
    with pytest.raises(RuntimeError, match="q_scale"):   # torch's own "Missing key(s)" report
        u2.load_state_dict(without, strict=True)
    missing, unexpected = u2.load_state_dict(without, strict=False)
    assert u2.attn_qk_norm == 2 and missing and all(k.endswith(("q_scale", "k_scale")) for k in missing) and not unexpected


def test_clone_of_a_switched_unet_still_follows_checkpoint_keys():
    """cast_model_parameters clones through the CONSTRUCTOR's kwargs: a mode a checkpoint switched on is carried over as
    the live mode, not as an explicit request, so the clone loads a checkpoint without q_scale / k_scale strictly and
    falls back to what the user built (cosine-sim here)."""
    from imagen_pytorch import Unet

    kw = dict(H.UNET_KW["small1"], cond_on_text=False, text_embed_dim=None)
    with_scales = R.Unet(**kw, attn_qk_norm=2).state_dict()
    u = Unet(**kw, cosine_sim_attn=True)
    u.load_state_dict(with_scales, strict=True)
    assert u.attn_qk_norm == 2 and u._locals["attn_qk_norm"] is None
    clone = u.cast_model_parameters(lowres_cond=True, text_embed_dim=None, channels=3, channels_out=3, cond_on_text=False)
    assert clone is not u and clone.attn_qk_norm == 2 and not clone._ctor_qk_explicit and clone._ctor_qk_norm == 1
    assert any(k.endswith("q_scale") for k in clone.state_dict())
    without = R.Unet(**kw, lowres_cond=True).state_dict()
    clone.load_state_dict(without, strict=True)
    assert clone.attn_qk_norm == 1


def test_restore_parts_reports_what_it_could_not_load_and_unets_deepcopy():
    """A checkpoint from another library version must not be half-loaded silently (sample_ultra_res.py:59-63
    falls back to restore_parts on any RuntimeError of the strict load)."""
    import copy

    from imagen_pytorch import Unet, restore_parts

    u = Unet(**H.UNET_KW["small1"], cond_on_text=False, text_embed_dim=None)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    dropped = "mid_block1.block1.project.bias"
    del sd[dropped]
    sd["mid_attn.layers.0.0.q_scale"] = torch.ones(64)
    sd["final_conv.bias"] = torch.ones(5)
    lines = []
    restore_parts(u.state_dict(), sd, report=lines.append)
    text = "\n".join(lines)
    assert "final_conv.bias" in text and dropped in text and "q_scale" in text and "qk-norm" in text
    lines.clear()
    restore_parts(u.state_dict(), u.state_dict(), report=lines.append)
    assert not lines                                    # a complete, matching checkpoint reports nothing
    u._engines = {"fake": 1}                            # stands for live ctypes plan handles
    c = copy.deepcopy(u)
    assert c._engines == {} and c._io_buffers == {} and u._engines == {"fake": 1}
    u._engines = {}
    for (ka, va), (kb, vb) in zip(u.state_dict().items(), c.state_dict().items()):
        assert ka == kb and torch.equal(va, vb) and va.data_ptr() != vb.data_ptr()


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    import imagen_pytorch as ip
    from imagen_pytorch import _engine as E

    u = ip.Unet(**H.UNET_KW["small1"], cond_on_text=False, text_embed_dim=None)
    if not torch.cuda.is_available():
        with pytest.raises(E.EngineUnavailable):
            u(torch.zeros(1, 3, 16, 16), torch.zeros(1))
        im = ip.Imagen([u], image_sizes=(16,), timesteps=2, condition_on_text=False)
        with pytest.raises(E.EngineUnavailable):
            im.sample(batch_size=1)
    with pytest.raises(RuntimeError):
        u.mid_block1(torch.zeros(1))  # parameter containers hold no arithmetic
    for f in (ROOT / "kidney-diffusion_amd").rglob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"
    for f in list((ROOT / "kidney-diffusion_amd" / "csrc").glob("*")):
        if f.is_file():
            assert "oracle" not in f.read_text(errors="ignore").lower() or f.name == "Makefile"


# ------------------------------------------------------------------------------- C ABI
def test_library_loads_and_exports_every_symbol_the_header_declares():
    from imagen_pytorch import _engine as E

    header = (ROOT / "include" / "kd_engine.h").read_text()
    declared = set(re.findall(r"\b(kd_[a-z0-9_]+)\s*\(", header))
    declared -= {"kd_sampler_create"}  # mentioned in prose only
    assert {"kd_unet_create", "kd_unet_forward", "kd_sample_loop", "kd_sample_steps", "kd_conv2d_nhwc",
            "kd_quantile_abs", "kd_philox_normal"} <= declared
    lib = E.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libkd_engine.so does not export {sym}"
    assert set(E.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.kd_version() == 2   # KD_ENGINE_ABI_VERSION of include/kd_engine.h
    assert C.sizeof(E.kd_unet_config_t) == 4 * (2 + 4 * E.KD_MAX_LEVELS + 27)  # ints only, header order
    assert lib.kd_quantile_workspace_bytes(4) == 4 * 16 + 4 * 4 * 256 * 4


def test_library_build_id_matches_the_sources_and_the_build_is_up_to_date():
    """The binary the tests load must be the one HEAD's sources build: kd_build_id() (sha256 of csrc/ + include/
    baked in at compile time) equals the hash of the sources on disk, `make -q` has nothing to do, and a
    library of another id is refused by the binding."""
    import shutil
    import subprocess

    from imagen_pytorch import _engine as E

    lib = E.load()
    have = lib.kd_build_id().decode()
    assert have.split("+")[0] == E.source_build_id(), "libkd_engine.so is stale: rebuild (make -C kidney-diffusion_amd/csrc)"

    class Fake:
        def kd_build_id(self):
            return b"0123456789abcdef"

    with pytest.raises(E.EngineUnavailable, match="built from other sources"):
        E._check_build_id(Fake(), "fake.so")
    if shutil.which("make") and Path("/opt/rocm/bin/hipcc").exists():
        csrc = ROOT / "kidney-diffusion_amd" / "csrc"
        assert subprocess.run(["make", "-C", str(csrc), "-q"]).returncode == 0, "objects older than their sources"


def test_makefile_tracks_header_dependencies(tmp_path):
    """Round 2 shipped a library that predated its last header edit: the Makefile listed headers by hand and
    missed epilogue.h.  Dependencies now come from the compiler (-MMD): touching epilogue.h must schedule
    exactly its users (kernels_conv, kernels_gemm_bf16x3, kernels_init, kernels_norm) + the link, and nothing when nothing changed."""
    import os
    import shutil
    import subprocess

    csrc = ROOT / "kidney-diffusion_amd" / "csrc"
    build = ROOT / "kidney-diffusion_amd" / "build"
    if not shutil.which("make") or not (build / "kernels_conv.d").exists():
        pytest.skip("no in-tree build with dependency files")
    hdr = csrc / "epilogue.h"
    st = hdr.stat()

    def planned():
        out = subprocess.run(["make", "-C", str(csrc), "-n"], capture_output=True, text=True, check=True).stdout
        return sorted(set(re.findall(r"-c (\S+\.hip)", out)))

    assert planned() == []
    try:
        os.utime(hdr, None)
        assert planned() == ["kernels_conv.hip", "kernels_gemm_bf16x3.hip", "kernels_init.hip", "kernels_norm.hip"]
    finally:
        os.utime(hdr, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert planned() == []


def test_bench_launches_its_own_ranks_when_called_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` is how the driver may call the benchmark: without WORLD_SIZE it must start N ranks itself
    (torch.distributed.run as a CHILD process, before any GPU call, rendezvous on 127.0.0.1), hand them its own arguments
    and exit with their code; under a launcher (WORLD_SIZE set) it must not spawn."""
    import importlib
    import subprocess
    import sys

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    calls = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one rank needs no launcher: main() goes on (and stops at the first GPU call on this box)
    calls.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            bench.main()
    assert not calls


def test_philox_host_reference_is_standard_normal_and_keyed():
    from oracle.philox_ref import philox4x32_10, philox_normal

    # Random123 known-answer vector for Philox4x32-10 (counter = key = 0)
    r = philox4x32_10(np.zeros(1), np.zeros(1), np.zeros(1), np.zeros(1), 0, 0)
    assert [int(x[0]) for x in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    a = philox_normal(1 << 16, 1, 2)
    b = philox_normal(1 << 16, 1, 3)
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1) < 0.02 and abs(np.mean(a * b)) < 0.02
    assert np.array_equal(a, philox_normal(1 << 16, 1, 2))
    assert np.array_equal(a[:1001], philox_normal(1001, 1, 2))  # prefix-stable


def test_fp32_a_loader_waves_keep_their_hand_placed_waits(tmp_path):
    """gemm_bf16x3_kernel<true, *>: the loader waves' A loads are inline assembly and their `s_waitcnt vmcnt` are placed by
    hand (kernels_gemm_bf16x3.hip: hipcc's own bookkeeping made every use wait for all outstanding accesses).  Two things the
    compiler must not do to that code, checked in the assembly it emits: copy a register whose load is still in flight (it
    did, in front of the waits, when the waits of the tail cases were separate statements - wrong results), or add waits of
    its own for the loaded registers."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    csrc = ROOT / "kidney-diffusion_amd" / "csrc"
    out = tmp_path / "x3.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
                        str(csrc / "kernels_gemm_bf16x3.hip"), f"-I{csrc}", f"-I{ROOT / 'include'}", "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    checked = 0
    for m in re.finditer(r"^(_ZN2kd18gemm_bf16x3_kernelILb1E\S+):[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.S | re.M):
        ins = [l.split(";")[0].strip() for l in m.group(2).split("\n")]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        dma = [i for i, l in enumerate(ins) if l.startswith("buffer_load_dwordx4") and l.endswith(" lds")]
        # the loader waves' code: from the first register load that shares the DMAs' neighbourhood to the last DMA
        loads = [i for i, l in enumerate(ins) if re.match(r"buffer_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\], s\d+ offen$", l)
                 and dma and dma[0] - 400 < i < dma[-1]]
        assert dma and len(loads) >= 16, (m.group(1), len(dma), len(loads))
        region = ins[loads[0]:dma[-1] + 40]
        assert not [l for l in region if re.match(r"v_mov_b(32|64)(_e32)? v\[?\d+[:\d\]]*, v", l)], m.group(1)
        waits = [l for l in region if l.startswith("s_waitcnt") and "vmcnt" in l]
        counts = sorted({int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in waits})
        assert counts == [0, 14, 17], (m.group(1), counts)   # the tail's plain wait, B pieces landed, A values landed
        checked += 1
    assert checked == 3   # epilogue kinds 0, 1, 2


def test_kernels_with_counted_vmcnt_waits_use_no_scratch(tmp_path):
    """conv_buf_kernel and wino_fused_kernel pace their LDS-DMA pipelines with counted `s_waitcnt vmcnt(N)`.
    A register spill would add scratch loads / stores to the same counter and let a barrier pass before the
    DMA data has landed (seen once in an experiment: wrong results AND 3x slower).  hipcc reports spills
    only on request, so ask: every kernel of these files must have ScratchSize 0 and no spilled VGPRs."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    csrc = ROOT / "kidney-diffusion_amd" / "csrc"
    srcs = ("kernels_conv.hip", "kernels_wino_fused128.hip", "kernels_init.hip", "kernels_gemm_bf16x3.hip")

    def compile_one(src):
        return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", str(csrc / src),
                               f"-I{csrc}", f"-I{ROOT / 'include'}", "-Rpass-analysis=kernel-resource-usage",
                               "-o", str(tmp_path / (src + ".o"))], capture_output=True, text=True, timeout=900)

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=4) as pool:   # the files compile side by side (about a minute each)
        outs = list(pool.map(compile_one, srcs))
    for src, out in zip(srcs, outs):
        assert out.returncode == 0, out.stderr[-2000:]
        scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
        spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", out.stderr)]
        assert scratch and len(scratch) == len(spills), f"{src}: no resource-usage remarks"
        assert not any(scratch) and not any(spills), f"{src}: scratch {scratch}, spills {spills}"
