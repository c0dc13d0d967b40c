"""ctypes binding of ``libkd_engine.so`` (C ABI: ``include/kd_engine.h``).

There is no CPU path: if the shared library is missing or no HIP device is visible, every entry
point raises ``EngineUnavailable`` — the package never falls back to PyTorch arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

KD_MAX_LEVELS = 8

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libkd_engine.so"


class EngineUnavailable(RuntimeError):
    pass


class EngineError(RuntimeError):
    pass


class kd_unet_config_t(C.Structure):
    _fields_ = [
        ("dim", C.c_int),
        ("num_levels", C.c_int),
        ("dim_mults", C.c_int * KD_MAX_LEVELS),
        ("num_resnet_blocks", C.c_int * KD_MAX_LEVELS),
        ("layer_attns", C.c_int * KD_MAX_LEVELS),
        ("layer_cross_attns", C.c_int * KD_MAX_LEVELS),
        ("cond_dim", C.c_int),
        ("channels", C.c_int),
        ("cond_images_channels", C.c_int),
        ("lowres_cond", C.c_int),
        ("memory_efficient", C.c_int),
        ("init_conv_to_final_conv_residual", C.c_int),
        ("cond_on_text", C.c_int),
        ("text_tokens", C.c_int),
        ("attn_heads", C.c_int),
        ("attn_dim_head", C.c_int),
        ("ff_mult_x2", C.c_int),
        ("num_time_tokens", C.c_int),
        ("sinu_dim", C.c_int),
        ("resnet_groups", C.c_int),
        ("attend_at_middle", C.c_int),
        ("use_gca", C.c_int),
        ("batch", C.c_int),
        ("image_size", C.c_int),
        ("conv_algo", C.c_int),
        ("attn_qk_norm", C.c_int),
        ("downsample_conv4", C.c_int),
        ("mid_attn_plain", C.c_int),
        ("wino_slice_mb", C.c_int),
        ("wino43_min_cin", C.c_int),
        ("gemm_bf16x3", C.c_int),
        ("x3_linear", C.c_int),
        ("wino4_max_images", C.c_int),
    ]


class kd_param_t(C.Structure):
    _fields_ = [("name", C.c_char_p), ("d_data", C.c_void_p), ("numel", C.c_int64)]


class kd_schedule_t(C.Structure):
    _fields_ = [("T", C.c_int)] + [
        (n, C.POINTER(C.c_float))
        for n in ("log_snr", "alpha", "sigma", "alpha_next", "sigma_next", "c", "noise_scale", "rn_a", "rn_b")
    ]


class kd_sample_args_t(C.Structure):
    _fields_ = [
        ("objective", C.c_int),
        ("dynamic_threshold", C.c_int),
        ("percentile", C.c_float),
        ("resample_times", C.c_int),
        ("d_lowres", C.c_void_p),
        ("d_lowres_log_snr", C.c_void_p),
        ("d_cond_images", C.c_void_p),
        ("d_text_tokens", C.c_void_p),
        ("d_text_hiddens", C.c_void_p),
        ("d_inpaint_images", C.c_void_p),
        ("d_inpaint_masks", C.c_void_p),
        ("d_noise_step", C.c_void_p),
        ("d_noise_inpaint", C.c_void_p),
        ("d_noise_renoise", C.c_void_p),
        ("seed", C.c_uint64),
        ("use_graph", C.c_int),
        ("cond_scale", C.c_float),
        ("d_null_text_tokens", C.c_void_p),
        ("d_null_text_hiddens", C.c_void_p),
        ("lowres_log_snr_uniform", C.c_int),
        ("lowres_log_snr_value", C.c_float),
        ("cond_table", C.c_int),
        ("cond_table_max_mb", C.c_int),
    ]


# symbol -> (restype, argtypes); tests/test_cpu.py::test_library_loads_and_exports_every_symbol_the_header_declares checks it against include/kd_engine.h
SIGNATURES = {
    "kd_last_error": (C.c_char_p, []),
    "kd_version": (C.c_int, []),
    "kd_build_id": (C.c_char_p, []),
    "kd_unet_create": (C.c_int, [C.POINTER(kd_unet_config_t), C.POINTER(kd_param_t), C.c_int,
                                 C.POINTER(C.c_void_p)]),
    "kd_unet_destroy": (None, [C.c_void_p]),
    "kd_unet_hbm_bytes": (C.c_int64, [C.c_void_p]),
    "kd_unet_create_shared": (C.c_int, [C.POINTER(kd_unet_config_t), C.POINTER(kd_param_t), C.c_int, C.c_void_p,
                                        C.POINTER(C.c_void_p)]),
    "kd_unet_weight_bytes": (C.c_int64, [C.c_void_p]),
    "kd_unet_macs": (C.c_int64, [C.c_void_p]),
    "kd_unet_mfma_macs": (C.c_int64, [C.c_void_p]),
    "kd_unet_mfma_bf16_macs": (C.c_int64, [C.c_void_p]),
    "kd_unet_num_launches": (C.c_int, [C.c_void_p]),
    "kd_unet_num_cond_launches": (C.c_int, [C.c_void_p]),
    "kd_unet_cond_table_build_ms": (C.c_float, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kd_unet_cond_table_refused_bytes": (C.c_int64, [C.c_void_p]),
    "kd_unet_text_cond": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "kd_unet_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_void_p]),
    "kd_unet_forward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8 + [C.c_void_p]),
    "kd_sample_loop": (C.c_int, [C.c_void_p, C.POINTER(kd_schedule_t), C.POINTER(kd_sample_args_t), C.c_void_p,
                                 C.c_void_p]),
    "kd_sample_steps": (C.c_int, [C.c_void_p, C.POINTER(kd_schedule_t), C.POINTER(kd_sample_args_t), C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p]),
    "kd_sample_build_cond_table": (C.c_int, [C.c_void_p, C.POINTER(kd_schedule_t), C.POINTER(kd_sample_args_t), C.c_int,
                                             C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "kd_sample_finalize": (C.c_int, [C.c_void_p, C.POINTER(kd_sample_args_t), C.c_void_p, C.c_void_p]),
    "kd_sample_last": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kd_conv2d_nhwc": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_void_p]),
    "kd_conv3x3_winograd_nhwc": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "kd_conv3x3_winograd4_nhwc": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "kd_gemm_bf16x3": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]),
    "kd_linear_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "kd_linear_bf16x3_seg_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "kd_downsample_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "kd_gn_conv3x3_winograd_fused_nhwc": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "kd_init_conv_nchw": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    "kd_groupnorm_silu_nhwc": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "kd_layernorm": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "kd_layernorm_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kd_layernorm_linear_bf16x3": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 3 +
                                   [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "kd_attention": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "kd_quantile_abs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t,
                                  C.c_void_p]),
    "kd_quantile_workspace_bytes": (C.c_size_t, [C.c_int]),
    "kd_cfg_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "kd_philox_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
}

_lib = None
ABI_VERSION = 2   # KD_ENGINE_ABI_VERSION of include/kd_engine.h


def lib_path() -> Path:
    return Path(os.environ.get("KD_ENGINE_LIB", _LIB_PATH))


def load() -> C.CDLL:
    """Loads the shared library (no GPU needed for loading/symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise EngineUnavailable(
            f"{path} not found — build it with `make -C kidney-diffusion_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.kd_version() != ABI_VERSION:   # structs and argument lists of this binding follow ONE header version
        raise EngineUnavailable(f"{path} speaks ABI version {lib.kd_version()}, this binding {ABI_VERSION} "
                                "(KD_ENGINE_ABI_VERSION of include/kd_engine.h)")
    _check_build_id(lib, path)
    _lib = lib
    return lib


def source_build_id():
    """Hash of the sources next to the package (csrc/build_id.py), or None when they are not there."""
    script = Path(__file__).resolve().parent.parent / "csrc" / "build_id.py"
    if not script.exists():
        return None
    import importlib.util

    spec = importlib.util.spec_from_file_location("_kd_build_id", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_id()


def _check_build_id(lib, path):
    """A library built from other sources than the ones lying next to it is refused (a stale binary once
    passed the GPU tests for a commit it did not contain).  KD_ENGINE_LIB (an explicit foreign library) and
    a source-less install skip the comparison."""
    if "KD_ENGINE_LIB" in os.environ:
        return
    want = source_build_id()
    if want is None:
        return
    have = lib.kd_build_id().decode()
    if "+" in have:
        raise EngineUnavailable(
            f"{path} is an experiment build ({have}: its kernels and plans follow KD_* environment variables); the product "
            "path loads it only when KD_ENGINE_LIB names it explicitly. Rebuild with `make -C kidney-diffusion_amd/csrc`.")
    if have != want:
        raise EngineUnavailable(
            f"{path} was built from other sources (library build id {have}, sources {want}): "
            "rebuild with `make -C kidney-diffusion_amd/csrc`.")


def check(rc: int):
    if rc != 0:
        raise EngineError(load().kd_last_error().decode("utf-8", "replace"))


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise EngineUnavailable("no HIP device visible: the MI355X engine has no CPU fallback")


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    import torch

    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "engine tensors: contiguous fp32 on device"
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
