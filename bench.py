#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the 64->256 super-res UNet (train_ultra_res.py:39-48
kwargs, 3 cond-image channels, low-res conditioned) at batch 16 on N MI355X (BASELINE.json).

One "step" = one (t -> t_next) iteration of the DDPM sampling loop for the whole batch: inpaint-free,
cond_scale 1 => exactly one UNet forward + x0/quantile/posterior/noise kernels.
Weights are random-init (no checkpoints exist offline), inputs synthetic; everything is resident in
HBM before the timed region.  N > 1: one process per GPU, every rank runs its own batch of 16 (the
path has no cross-sample exchange: replicas of independent batches, weak scaling, no collective in
the data path); rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # without WORLD_SIZE: spawns exactly that launcher itself (one child
                                          # process per GPU, as the reference spawns its workers,
                                          # sample_ultra_res.py:235-240) before any GPU call and relays the line

The line of the headline workload also carries `rccl_ranks` (an all-reduce of ones over the process group: the
number of ranks the collective library saw) and a nested `grid` object: patches/s of the 8x8 ultra-res grid
(BASELINE configs[4], the >= 6x scaling target) for 1 and 3 canvases in flight, measured by the same command.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "kidney-diffusion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402

BATCH, SIZE, T_SCHED = 16, 256, 250
DOMINANT = ("wino_fused_gn128_kernel: fused Winograd F(2x2,3x3) 3x3 convs of the ResnetBlocks, GroupNorm/FiLM/SiLU applied "
            "in the kernel (sixteen waves, persistent workgroups, items of 16x8 pixels x 128 output channels)")
WINO4 = ("conv_buf_kernel: the 36 position GEMMs of Winograd F(4x4,3x3) - the ResnetBlock 3x3 convs with Cin >= 512 "
         "(buffer-DMA implicit-GEMM kernel, batched over the positions)")
WINO4_X3 = ("gemm_bf16x3_kernel: the 36 position GEMMs of Winograd F(4x4,3x3) - the ResnetBlock 3x3 convs, Cin >= 128 - as fp32 "
            "products on the bf16 matrix pipe (three bf16 pieces per fp32 operand, six exact products per k-step, fp32 "
            "accumulation; persistent 256 x 128 tiles, LDS-DMA ring fed by loader waves; V as planes, or as fp32 split by the "
            "loader waves where the GEMM waits for HBM: the Cout = 128 layers)")
LIN_X3 = ("gemm_bf16x3_kernel, epilogue form: attention projections, feed-forward, 1x1 skip convs, upsample and 2x2-s2 "
          "downsample convs with K >= 256 as fp32 products on the bf16 matrix pipe (fp32 activations split by the kernel's loader waves; bias / "
          "residual / GlobalContext gate / SiLU + PixelShuffle epilogue, GroupNorm partials of the output)")
X3_ALL = ("gemm_bf16x3_kernel, all launches of the step (the two entries above without their sum_slabs_kernel launches): fp32 "
          "products on the bf16 matrix pipe - three bf16 pieces per fp32 operand, six exact products per k-step, fp32 accumulation; "
          "persistent 256 x 128 tiles, 8 computing + 4 loader waves, 4-stage LDS-DMA ring")
CONV_CLASS = "conv_buf_kernel / conv_igemm_kernel / init_conv_kernel: 1x1, 2x2-s2, init and final convs, token GEMMs"
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix = vector peak (v_mfma_f32_32x32x2_f32, exact fp32)
# dense bf16 matrix peak (MI355X_MICROARCH.md "~2.5 PF dense": v_mfma_f32_32x32x16_bf16 at 32 cycles per SIMD, 1024 CU-SIMDs,
# 2.4 GHz).  The bf16x3 GEMMs (kernels_gemm_bf16x3.hip) issue SIX bf16 MACs per fp32 MAC: their fp32-equivalent ceiling is
# a sixth of this, 419 TFLOP/s
BF16_PEAK_TFLOPS = 2516.6
SR_UNET_KW = dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=2, memory_efficient=True,
                  layer_attns=(False, False, False, True), layer_cross_attns=(False, False, True, True),
                  init_conv_to_final_conv_residual=True, cond_images_channels=3)  # train_ultra_res.py:39-48


def build_unet(seed=0):
    import imagen_pytorch as ip

    torch.manual_seed(seed)
    u = ip.Unet(**SR_UNET_KW, lowres_cond=True, cond_on_text=False, text_embed_dim=None)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():  # the library zero-inits final_conv; re-init so outputs are non-trivial
        u.final_conv.weight.copy_(torch.randn(u.final_conv.weight.shape, generator=g) * 0.02)
        u.final_conv.bias.copy_(torch.randn(u.final_conv.bias.shape, generator=g) * 0.02)
    return u


def synthetic_inputs(batch, device=None, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, SIZE, SIZE, generator=g)
    start = torch.rand(batch, 3, 64, 64, generator=g)
    cond = torch.rand(batch, 3, SIZE, SIZE, generator=g)  # already at the UNet's resolution
    lowres = torch.nn.functional.interpolate(start, SIZE, mode="nearest") * 2 - 1
    lowres_noise = torch.randn(batch, 3, SIZE, SIZE, generator=g)
    mv = (lambda t: t.to(device).contiguous()) if device is not None else (lambda t: t)
    return mv(x), mv(lowres), mv(lowres_noise), mv(cond)


def kernel_classes(lib, handle, iters=3):
    """Per-kernel-class rates of one UNet forward, measured live with HIP events around every launch
    (`kd_unet_profile`, include/kd_engine.h), on the stream the launches run on.  For a matrix-core class
    `achieved` = the FLOPs its launches ISSUE on the MFMA pipe (the engine's own per-op count: Winograd
    F(2x2,3x3) layers issue 16/36 of the direct convolution's MACs, K padded to the kernel's chunk) divided by
    their summed device time, so `frac` = achieved / peak <= 1 is a pipe utilisation; `achieved_direct_equiv`
    prices the same time against the direct-convolution FLOPs the reference computes."""
    import re

    from imagen_pytorch import _engine as E

    buf = C.create_string_buffer(1 << 20)
    E.check(lib.kd_unet_profile(handle, iters, buf, len(buf), E.current_stream()))
    rows = [l.split(",") for l in buf.value.decode().strip().split("\n")[1:]]
    cls = {}

    def add(key, us, flop=0.0, issued=0.0, nbytes=0.0):
        c = cls.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
        c[0] += 1
        c[1] += us
        c[2] += flop
        c[3] += issued
        c[4] += nbytes

    def hbm_bytes(label):
        """Algorithmic HBM bytes of an HBM-bound launch from its label (every map read / written once; fp32)."""
        g = lambda pat: [int(v) for v in re.match(pat, label).groups()] if re.match(pat, label) else None
        v = g(r"gate_add HW(\d+) C(\d+)")
        if v: return 3 * 4.0 * BATCH * v[0] * v[1]                  # h2, x in; out
        v = g(r"gca_pool HW(\d+) C(\d+)")
        if v: return 4.0 * BATCH * v[0] * v[1]                      # one read of the map
        v = g(r"gn stats HW(\d+) C(\d+)")
        if v: return 4.0 * BATCH * v[0] * v[1]
        v = g(r"gn apply HW(\d+) C(\d+)")
        if v: return 2 * 4.0 * BATCH * v[0] * v[1]
        v = g(r"ln rows(\d+) C(\d+)")
        if v: return 2 * 4.0 * v[0] * v[1]
        v = g(r"ln x2 rows(\d+) C(\d+)")
        if v: return 4 * 4.0 * v[0] * v[1]                         # x and the residual in; both LayerNorms' rows out
        v = g(r"(?:scale slice|skip copy|concat head|concat tail|concat) rows(\d+) C(\d+)")
        if v: return 2 * 4.0 * v[0] * v[1]
        v = g(r"attn N(\d+)")
        if v: return 4.0 * BATCH * v[0] * (2 * 512 + 2 * 64)      # q, o: 8 heads x 64; k, v: one shared head
        v = g(r"xattn Nq(\d+) Nk(\d+)")
        if v: return 4.0 * BATCH * (2 * v[0] * 512 + 2 * v[1] * 512)
        if label.startswith("final gather"): return 4.0 * BATCH * SIZE * SIZE * (32 + 3 + 3)
        return 0.0   # GroupNorm folds, tiny element-wise launches: latency, not bytes

    total_us = 0.0
    sums_us = {}   # time of sum_slabs_kernel launches booked to a GEMM class
    x3_parts = {"position GEMMs, V as fp32 (HBM side)": [], "position GEMMs, V as planes (matrix side)": [],
                "token GEMMs / 1x1 convs": []}   # (us, bf16 FLOP issued, algorithmic bytes) per gemm_bf16x3_kernel launch
    for _, label, macs, us, mfma in rows:
        us, macs, mfma = float(us), int(macs), int(mfma)
        total_us += us
        m = re.match(r"(wino_in|wino_out|wino gemm|wino4_in3|wino4_in|wino4_out|wino4 gemm bf16x3|wino4 gemm) M(\d+) Cin(\d+) Cout(\d+)", label)
        if re.match(r"conv k[12] x3 sum", label):   # the k-parts of the tiles added and the epilogue applied: time of the same GEMMs
            cls[LIN_X3][1] += us
            sums_us[LIN_X3] = sums_us.get(LIN_X3, 0.0) + us
        elif re.match(r"conv k[12] x3", label):     # `mfma` = bf16 MACs (6 per fp32 MAC)
            add(LIN_X3, us, 2.0 * macs, 2.0 * mfma)
            g3 = re.match(r"conv k[12] x3 M(\d+) Cin(\d+) Cout(\d+)", label)
            if g3:   # rows x (K in, fp32 - planes where a LayerNorm wrote them: priced as fp32 here - + N out) + the weights' planes
                rows, kin, nout = (int(v) for v in g3.groups())
                kin *= 4 if label.startswith("conv k2") else 1
                x3_parts["token GEMMs / 1x1 convs"].append((us, 2.0 * mfma, 4.0 * rows * (kin + nout) + 6.0 * kin * nout))
        elif label.startswith("conv k3"):
            add("conv_buf_kernel: direct 3x3 convs", us, 2.0 * macs, 2.0 * mfma)
        elif label.startswith("wino fused"):
            add(DOMINANT, us, 2.0 * macs, 2.0 * mfma)
        elif m and m.group(1) == "wino gemm":
            add("conv_buf_kernel: Winograd F(2x2,3x3) position GEMMs", us, 2.0 * macs, 2.0 * mfma)
        elif m and m.group(1) == "wino4 gemm bf16x3":   # `mfma` = the bf16 MACs (6 per fp32 MAC of the 36 GEMMs)
            add(WINO4_X3, us, 2.0 * macs, 2.0 * mfma)
            # V and D of the 36 positions: 2.25 values per pixel and channel; V 4 B as fp32 (the plan's rule: where
            # Cin Cout / (6 Cin + 4 Cout) < 40), 6 B as planes; D 4 B; the weights' planes once
            px, cin, cout = int(m.group(2)), int(m.group(3)), int(m.group(4))
            f32v = cin * cout < 40 * (6 * cin + 4 * cout)
            x3_parts["position GEMMs, V as fp32 (HBM side)" if f32v else "position GEMMs, V as planes (matrix side)"].append(
                (us, 2.0 * mfma, 2.25 * px * (cin * (4.0 if f32v else 6.0) + cout * 4.0) + 36 * 6.0 * cin * cout))
        elif label.startswith("wino4 x3 sum"):   # the left-over tiles' k-parts added up: time of the same GEMMs (not a launch of the class's count)
            cls[WINO4_X3][1] += us
            sums_us[WINO4_X3] = sums_us.get(WINO4_X3, 0.0) + us
        elif m and m.group(1) == "wino4 gemm":
            add(WINO4, us, 2.0 * macs, 2.0 * mfma)
        elif m and m.group(1).startswith("wino4"):
            # per pixel and channel: the input transform reads the map once (4 B) and writes V, 2.25 values: 9 B as fp32,
            # 13.5 B as the three bf16 planes of the bf16x3 GEMM (6 B per value); the output transform reads D (9 B) and
            # writes the map (4 B)
            ch = int(m.group(4)) if m.group(1) == "wino4_out" else int(m.group(3))
            per = {"wino4_in": 13.0, "wino4_in3": 17.5, "wino4_out": 13.0}[m.group(1)]
            add("wino4_in_kernel + wino4_out_kernel (Winograd F(4x4,3x3) transforms, GroupNorm/FiLM/SiLU and statistics fused)", us,
                nbytes=per * int(m.group(2)) * ch)
        elif m:  # transforms move 5x the map: read 1x / write 4x (in), read 4x / write 1x (out)
            ch = int(m.group(3)) if m.group(1) == "wino_in" else int(m.group(4))
            add("wino_in_kernel + wino_out_kernel (Winograd transforms)", us, nbytes=20.0 * int(m.group(2)) * ch)
        elif label.startswith("skinny"):   # M <= 32 rows: the weights are read once, nothing is reused
            mk = re.match(r"skinny M(\d+) K(\d+) N(\d+)", label)
            add("linear_skinny_mfma_kernel (time MLPs, GlobalContext FCs: weight-bandwidth / latency bound)", us,
                nbytes=4.0 * int(mk.group(2)) * int(mk.group(3)) if mk else 0.0)
        elif label.startswith("conv") or label.startswith("init conv"):
            add(CONV_CLASS, us, 2.0 * macs, 2.0 * (mfma or macs))
        else:
            add("GroupNorm, LayerNorm, attention core, GlobalContext, concat, gate (HBM-bound)", us, nbytes=hbm_bytes(label))
    out = []
    for key, (n, us, flop, issued, nbytes) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        e = {"kernel": key, "launches": n, "ms": us / 1e3, "avg_us": us / n, "share": us / total_us}
        if key in sums_us:   # the GEMM kernel's own launches, without the launches that add the k-parts
            ko = us - sums_us[key]
            e.update(kernel_only_ms=ko / 1e3, kernel_only_avg_us=ko / n, kernel_only_achieved=issued / ko / 1e6,
                     kernel_only_frac=issued / ko / 1e6 / BF16_PEAK_TFLOPS)
        if flop and key in (WINO4_X3, LIN_X3):
            e.update(bound="mfma", achieved=issued / us / 1e6, unit="TFLOP/s (bf16 MFMA)", peak=BF16_PEAK_TFLOPS,
                     frac=issued / us / 1e6 / BF16_PEAK_TFLOPS, achieved_fp32_equiv=issued / 6.0 / us / 1e6,
                     achieved_direct_equiv=flop / us / 1e6, issued_tflop_per_step=issued / 1e12)
        elif flop:
            e.update(bound="mfma", achieved=issued / us / 1e6, unit="TFLOP/s", peak=FP32_PEAK_TFLOPS,
                     frac=issued / us / 1e6 / FP32_PEAK_TFLOPS, achieved_direct_equiv=flop / us / 1e6,
                     issued_tflop_per_step=issued / 1e12)
        elif nbytes:
            e.update(bound="hbm", achieved=nbytes / us / 1e3, unit="GB/s", peak=8000.0, frac=nbytes / us / 1e3 / 8000.0)
        out.append(e)
    # gemm_bf16x3_kernel as ONE kernel (all its template instances: position GEMMs with V as planes / as fp32, token-GEMM and
    # 1x1-conv epilogue forms), without the sum_slabs_kernel launches: what rocprofv3's per-kernel table shows
    x3 = [(k, cls[k]) for k in (WINO4_X3, LIN_X3) if k in cls]
    if x3:
        n = sum(c[0] for _, c in x3)
        us = sum(c[1] - sums_us.get(k, 0.0) for k, c in x3)
        flop, issued = sum(c[2] for _, c in x3), sum(c[3] for _, c in x3)
        parts = [{"launches_of": k, "launches": len(v), "ms": sum(t[0] for t in v) / 1e3,
                  "mfma_frac": sum(t[1] for t in v) / sum(t[0] for t in v) / 1e6 / BF16_PEAK_TFLOPS,
                  "hbm_frac": sum(t[2] for t in v) / sum(t[0] for t in v) / 1e3 / 8000.0,
                  "algorithmic_gb": sum(t[2] for t in v) / 1e9} for k, v in x3_parts.items() if v]
        out.append({"kernel": X3_ALL, "combined": True, "launches": n, "ms": us / 1e3, "avg_us": us / n, "share": us / total_us,
                    "sum_launches_ms": sum(sums_us.values()) / 1e3,
                    "algorithmic_bytes_per_launch": sum(t[2] for v in x3_parts.values() for t in v) / n,
                    "by_side": parts,
                    "bound": "mfma", "achieved": issued / us / 1e6, "unit": "TFLOP/s (bf16 MFMA)", "peak": BF16_PEAK_TFLOPS,
                    "frac": issued / us / 1e6 / BF16_PEAK_TFLOPS, "achieved_fp32_equiv": issued / 6.0 / us / 1e6,
                    "achieved_direct_equiv": flop / us / 1e6, "issued_tflop_per_step": issued / 1e12})
    # the class VERDICT r4 item 1 tracks: every 1x1 / 2x2-s2 / init / final conv and token GEMM, whichever kernel runs it
    both = [e for e in out if e["kernel"] in (CONV_CLASS, LIN_X3)]
    if len(both) == 2:
        out.append({"kernel": "1x1 / 2x2-s2 / init / final convs and token GEMMs, both kernels together (the two entries above)",
                    "launches": sum(e["launches"] for e in both), "ms": sum(e["ms"] for e in both),
                    "share": sum(e["share"] for e in both),
                    "achieved_fp32_equiv": sum(e.get("achieved_fp32_equiv", e.get("achieved", 0.0)) * e["ms"] for e in both)
                    / sum(e["ms"] for e in both), "unit": "TFLOP/s of fp32-equivalent work"})
    return out


def host_cpu_info():
    """Threads the CPU baseline may use = min(physical cores, affinity mask, cgroup CPU quota) and the CPU model
    string.  torch's default (all logical CPUs of the host) oversubscribes a box that is given a CPU share."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    physical = len(cores) or max(1, logical // 2)
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except (OSError, ValueError):
            pass
    threads = min(x for x in (physical, affinity, quota) if x)
    return dict(model=model, logical=logical, physical=physical, affinity=affinity, cgroup_quota=quota, threads=threads)


PARITY_TOL = 2e-5   # rel-L2 of one UNet forward, engine vs CPU fp32 oracle (the tolerance the full-size tests state)
CPU_K0 = 125   # schedule index of the CPU baseline's first step: mid-schedule, where the UNet output drives x_(t-1)


def cpu_baseline(unet_product, timed_steps=3):
    """The oracle (CPU fp32 torch restatement of the reference's path) timed on this host AT THE HEADLINE BATCH (16): one
    warm-up step (also the parity reference: the engine's plan depends on the batch) and `timed_steps` timed steps of
    p_sample with injected noise - about 12 s each on the box's 16 threads.  Returns (json entry, inputs and outputs of
    the warm-up step for the parity check against the engine)."""
    from oracle import imagen_ref as R
    from oracle import sampler_ref as RS

    info = host_cpu_info()
    torch.set_num_threads(info["threads"])
    ou = R.Unet(**SR_UNET_KW, lowres_cond=True, cond_on_text=False, text_embed_dim=None)
    ou.load_state_dict(unet_product.state_dict(), strict=True)
    ou.eval()
    oim = RS.Imagen([R.NullUnet(), ou], image_sizes=(64, SIZE), timesteps=(T_SCHED, T_SCHED),
                    pred_objectives=("noise", "noise"), condition_on_text=False)
    x16, lowres16, lowres_noise16, cond16 = synthetic_inputs(BATCH)
    sched = oim.noise_schedulers[1]
    lowres16 = oim.lowres_noise_schedule.q_sample(lowres16, torch.full((BATCH,), 0.2), lowres_noise16)
    g = torch.Generator().manual_seed(99)
    preds = []
    fwd = ou.forward_with_cond_scale

    def fwd_keep(*a, **k):   # the UNet's output of every step (the parity check reads the warm-up's)
        out = fwd(*a, **k)
        preds.append(out)
        return out

    ou.forward_with_cond_scale = fwd_keep

    def step(xs, lr, cd, k):
        n = xs.shape[0]
        t, tn = sched.get_sampling_timesteps(n)[k]
        noise = torch.randn(xs.shape, generator=g)
        kw = dict(noise_scheduler=sched, text_embeds=None, text_mask=None, cond_images=cd, lowres_cond_img=lr,
                  lowres_noise_times=torch.full((n,), 0.2), cond_scale=1.0, pred_objective="noise", dynamic_threshold=True)
        t0 = time.perf_counter()
        x_next, x0 = oim.p_sample(ou, xs, t, noise, t_next=tn, **kw)
        return x_next, x0, noise, time.perf_counter() - t0

    dts = []
    with torch.no_grad():
        x_next, x0, noise, dt = step(x16, lowres16, cond16, CPU_K0)
        dts.append(dt)
        first = dict(x=x16, noise=noise, lowres=lowres16, cond=cond16, x_next=x_next, pred=preds[0], x0=x0)
        x = x_next
        for k in range(CPU_K0 + 1, CPU_K0 + 1 + timed_steps):
            x, _, _, dt = step(x, lowres16, cond16, k)
            dts.append(dt)
    timed = dts[1:]
    sec_per_step = sum(timed) / len(timed)
    quota = info["cgroup_quota"]
    entry = {"value": 1.0 / sec_per_step, "unit": "denoising-steps/s (batch 16)", "cores": info["threads"],
             "kind": "port",
             "cpu_model": info["model"],
             "host": {k: info[k] for k in ("logical", "physical", "affinity", "cgroup_quota")},
             "sample": f"oracle p_sample (UNet forward + x0 / dynamic threshold / posterior / noise) on CPU at the headline batch "
                       f"{BATCH}: 1 warm-up step ({dts[0]:.2f} s; the parity reference) + {len(timed)} timed steps "
                       f"({', '.join(f'{d:.2f}' for d in timed)} s), mean; torch {torch.__version__} oneDNN fp32, "
                       f"torch.set_num_threads({info['threads']}) = min(physical cores, affinity, cgroup quota): "
                       + (f"{info['threads']} threads = the CPU quota of this box's cgroup, on a host of {info['physical']} "
                          f"physical cores - NOT all host cores, which the box is not given"
                          if quota and quota < info["physical"] else f"all {info['physical']} physical cores of the host")}
    return entry, first


def engine_parity(unet, first, device, lib):
    """The engine's denoising step on the inputs of the CPU baseline's first step (same weights, same x_t,
    same conditioning, same injected noise, schedule index CPU_K0): relative L2 and max-abs of x_{t-1} against the
    oracle's, at the headline batch: the plan that was timed."""
    from imagen_pytorch import _engine as E
    from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes, beta_linear_log_snr

    b = first["x"].shape[0]
    h = unet.engine(b, SIZE, device, with_text=False)
    dv = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
    x, noise, lowres, cond = dv(first["x"]), dv(first["noise"]), dv(first["lowres"]), dv(first["cond"])
    ls_lr = dv(beta_linear_log_snr(torch.full((b,), 0.2)))
    tables = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=T_SCHED).step_tables()
    sc = E.kd_schedule_t()
    sc.T = 1                         # a one-step schedule holding the scalars of step CPU_K0
    keep = {name: v[CPU_K0:CPU_K0 + 1].contiguous() for name, v in tables.items()}
    for name, v in keep.items():
        setattr(sc, name, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))
    sa = E.kd_sample_args_t()
    sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times = 0, 1, 0.95, 1
    sa.d_lowres, sa.d_lowres_log_snr, sa.d_cond_images = E.ptr(lowres), E.ptr(ls_lr), E.ptr(cond)
    sa.lowres_log_snr_uniform, sa.lowres_log_snr_value = 1, float(ls_lr[0])
    sa.d_noise_step = E.ptr(noise)   # index 0 of the [T*R, B, 3, S, S] layout
    sa.use_graph = 1
    E.check(lib.kd_sample_steps(h, C.byref(sc), C.byref(sa), E.ptr(x), 0, 1, E.current_stream()))
    # what the step computed on the way (kd_sample_last): the UNet's eps-hat and the x0 estimate.  x_(t-1) itself is
    # ~97 % the identical input x_t and the identical injected noise at this step (c ~ 0.025), so an error of the UNet
    # would be attenuated ~50x in it: it is reported, but the asserted figures are those of eps-hat and x0-hat
    pred, x0raw, thr = torch.empty_like(x), torch.empty_like(x), torch.empty(b, device=device)
    for which, dst in ((0, pred), (1, x0raw), (2, thr)):
        E.check(lib.kd_sample_last(h, which, E.ptr(dst), E.current_stream()))
    torch.cuda.synchronize()
    sth = thr[:, None, None, None]
    x0 = torch.maximum(torch.minimum(x0raw, sth), -sth) / sth   # the clamp of ddpm_update_kernel, restated for the check

    def rel(got, ref):
        got, ref = got.double().cpu(), ref.double()
        return float((got - ref).norm() / ref.norm())

    out = {"parity_pred_rel_l2": rel(pred, first["pred"]), "parity_x0_rel_l2": rel(x0, first["x0"]),
           "parity_xprev_rel_l2": rel(x, first["x_next"]),
           "parity_pred_max_abs": float((pred.double().cpu() - first["pred"].double()).abs().max()),
           "parity_tolerance_rel_l2": PARITY_TOL,
           "parity_case": f"denoising step {CPU_K0} of {T_SCHED} at batch {b}: engine (default plan, Winograd convs) vs the CPU "
                          "oracle on identical weights / x_t / conditioning / injected noise: rel-L2 of the UNet output "
                          "eps-hat (pred), of the thresholded x0 estimate, and of x_(t-1) (the last is dominated by the shared "
                          f"input and noise); asserted: pred and x0 < {PARITY_TOL:g} (tests/test_fullsize_gpu.py)"}
    assert out["parity_pred_rel_l2"] < PARITY_TOL and out["parity_x0_rel_l2"] < PARITY_TOL, out
    return out


F_, T_ = False, True
ULTRA_UNETS = {  # train_ultra_res.py:29-60 (magnification level > 0: 3 conditioning channels)
    1: dict(dim=256, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
            layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=3),
    2: SR_UNET_KW,
    3: dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 6, 8), memory_efficient=True, layer_attns=False,
            layer_cross_attns=(F_, F_, F_, T_), init_conv_to_final_conv_residual=True, cond_images_channels=3),
}


def grid_workload(args, world, rank, device, distributed, barrier, canvases_list, T, steps, warmup):
    """BASELINE configs[4]: the 8x8 ultra-res outpainting grid of 1024-px patches through the 3-stage
    cascade (sample_ultra_res.py:264-448), sharded over the ranks by anti-diagonal waves with point-to-point
    strip exchange and one final all-gather (ultra_res/distributed.py).  One "step" = one full set of `canvases` grids.
    Strong scaling: the same 64 x canvases patches whatever N is.  Returns, on rank 0, one result
    dict per entry of canvases_list (the models are loaded and warmed once)."""
    import torch.distributed as dist

    import imagen_pytorch as ip
    from ultra_res import distributed as D
    from ultra_res import grid as G

    class FixedNullUnet(ip.NullUnet):  # train_ultra_res.py:65-75
        def __init__(self, lowres_cond=False, *a, **k):
            super().__init__()
            self.lowres_cond = lowres_cond
            self.dummy_parameter = torch.nn.Parameter(torch.tensor([0.0]))

        def cast_model_parameters(self, *a, **k):
            return self

        def forward(self, x, *a, **k):
            return x

    imagens = {}

    def load_imagen(stage):  # train_ultra_res.py:79-90 with one real UNet resident per Imagen
        # Built on the meta device and initialised ON THE GPU from a per-stage seed: every rank gets bit-identical
        # weights (patches sampled on different ranks must agree) without 8 ranks each drawing 2 x 10^9 numbers on
        # the host cores they share.  Matrices N(0, 1 / fan_in), norm gains 1, biases 0, final conv N(0, 0.02).
        with torch.device("meta"):
            unets = tuple(ip.Unet(**ULTRA_UNETS[i]) if i == stage else FixedNullUnet(lowres_cond=i > 1) for i in (1, 2, 3))
            im = ip.Imagen(unets=unets, image_sizes=(64, 256, 1024), timesteps=(T, T, T),
                           pred_objectives=("noise", "noise", "noise"), random_crop_sizes=(None, None, 256),
                           condition_on_text=False)
        im = im.to_empty(device=device)
        gen = torch.Generator(device=device).manual_seed(100 + stage)
        with torch.no_grad():
            for name, p_ in im.named_parameters():
                if "final_conv" in name:
                    p_.normal_(0, 0.02, generator=gen)
                elif p_.dim() > 1:
                    p_.normal_(0, float(p_[0].numel()) ** -0.5, generator=gen)
                elif name.endswith("weights"):             # LearnedSinusoidalPosEmb frequencies
                    p_.normal_(0, 1.0, generator=gen)
                elif name.endswith((".g", "weight")):      # 1-D: GroupNorm / LayerNorm gains
                    p_.fill_(1.0)
                else:                                      # biases, placeholders
                    p_.zero_()
        imagens[stage] = im
        return im

    geom = G.grid_geometry(1024, 1, 0.25)  # sample_ultra_res.py:280,307,311: 8x8 patches, canvas 6400
    assert geom.num_patches_width == 8 and geom.canvas_width == 6400
    n = args.grid_n
    pos = [(i, j) for i in range(n) for j in range(n)]
    g = torch.Generator().manual_seed(1234)
    zoomed = torch.rand(1, 3, 1024, 1024, generator=g).to(device)
    cond = G.cond_images_for_grid(zoomed, geom, pos)  # built in HBM: no per-patch host-to-device copy
    # --grid-batch N: same-wave patches of a rank share one sample() call in stages 1 and 2 (stage 3 is
    # already at 141 TFLOP/s per batch-1 patch and its workspace is 10 GB per sample)
    gb = {1: args.grid_batch, 2: args.grid_batch, 3: 1}
    sample_fn = D.imagen_sample_fn(load_imagen, args.grid_resample, device, use_graph=not args.no_graph, seed=1234,
                                   max_batch=gb)
    # finished patches live where the all-gather runs: HBM under RCCL, host memory in a gloo rehearsal
    # (gloo has no CUDA all_gather)
    slab_dev = device if (not distributed or dist.get_backend() == "nccl") else torch.device("cpu")

    # deal order inside a generalised wave: batch-1 step times per stage (profiles/README.md) x this run's timesteps
    stage_cost = {1: 6.9 * T, 2: 5.7 * T, 3: 43.0 * T}

    xstats = {}

    def run(positions, cond_images, canvases):
        out = D.sample_grids(sample_fn, (1, 2, 3), [positions] * canvases, [cond_images] * canvases, 0.25,
                             [n] * canvases, patch_width=geom.patch_width, device=slab_dev,
                             pipeline=not args.no_pipeline, stage_cost=stage_cost,
                             overlap_stages=None if (args.no_overlap or args.no_pipeline) else device,
                             gather=args.grid_gather, stats=xstats)
        if args.grid_gather == "root" and rank != 0:
            return None
        sub = G.GridGeometry(geom.patch_width, geom.patch_dist, n, geom.out_patch_dist,
                             1024 + (n - 1) * geom.out_patch_dist)
        return [G.stitch_canvas(o, positions, sub, background=zoomed.to(o[0].device)) for o in out]

    # Warm-up on EVERY rank: load the three stages, build every plan shape the timed run can use (batch 1..grid_batch
    # in stages 1-2, batch 1 in stage 3) and capture their step graphs, so none of that lands in the timed region
    # (one patch through run() would only exercise rank 0 at batch 1).
    for _ in range(max(1, warmup)):
        sample_fn.warm({st: range(1, gb[st] + 1) for st in (1, 2, 3)}, cond[0])

    if distributed:
        D.connect_ranks(None, slab_dev)   # RCCL send / recv connections between every pair, outside the timed region

    from imagen_pytorch import _engine as E

    lib = E.load()
    R = args.grid_resample
    flop_patch = issued_patch = 0.0
    for stage, size in ((1, 64), (2, 256), (3, 1024)):
        h = imagens[stage].unets[stage - 1].engine(1, size, device, with_text=False)
        flop_patch += 2.0 * lib.kd_unet_macs(h) * T * R         # direct convolutions, as the reference computes them
        issued_patch += 2.0 * lib.kd_unet_mfma_macs(h) * T * R  # what the plans put on the matrix cores (Winograd: 16/36)

    results = []
    for ncan in canvases_list:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            canv = run(pos, cond, ncan)
        barrier()
        elapsed = time.perf_counter() - t0
        if canv is not None:
            assert all(torch.isfinite(c).all() for c in canv)
            assert all(tuple(c.shape[-2:]) == (1024 + (n - 1) * geom.out_patch_dist,) * 2 for c in canv)
        per_rank = None
        if distributed:
            tmax = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            # what each rank did in the last grid of the timed region: so that a scaling curve explains itself
            mine = {k: xstats.get(k) for k in ("tasks_this_rank", "idle_waves_this_rank", "recv_wait_stream_s_this_rank",
                                               "recv_wait_host_s_this_rank", "p2p_bytes_sent_by_this_rank",
                                               "p2p_batches_posted_by_this_rank")}
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        patches = len(pos) * ncan * steps
        orient = [G.choose_orientation(pos)] * ncan
        waves = D.merged_waves([pos] * ncan, orient)
        achieved = patches * issued_patch / elapsed / 1e12 / world
        direct_equiv = patches * flop_patch / elapsed / 1e12 / world
        # what the schedule itself allows: generalised waves dealt by assign_tasks; a rank starts its share of a wave
        # when it has finished the previous one and the producers of that share are done - no barrier between the
        # waves (additive task costs, free exchange, no overlap of the light stages credited), relative to one rank
        def makespan(nranks):
            return D.ExchangePlan([pos] * ncan, orient, (1, 2, 3), nranks, not args.no_pipeline, stage_cost).makespan(stage_cost)
        schedule_bound = {str(k): makespan(1) / makespan(k) for k in (1, 2, 4, 8)}
        plan8 = {k: D.ExchangePlan([pos] * ncan, orient, (1, 2, 3), k, not args.no_pipeline, stage_cost) for k in (world, 8)}
        results.append({
            "metric": "patches/sec (ultra-res outpainting grid, 1024-px patches, 3-stage cascade)",
            "value": patches / elapsed, "unit": "patches/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed * 1e3 / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: {n}x{n} grid x {ncan} canvas(es), overlap 0.25, stages 64->256->1024 "
                                   f"(train_ultra_res.py:27-92), batch 1 per patch as the reference samples them, "
                                   f"timesteps ({T},{T},{T}) [reference default (1024,256,256)], inpaint_resample {R}, "
                                   f"stage-1/2 patches per sample() call <= {args.grid_batch}, random-init weights",
                       "canvases": ncan, "patches": len(pos) * ncan, "schedule_slots": D.schedule_length(waves, world),
                       "schedule_speedup_bound": schedule_bound,
                       "exchange": {
                           "p2p_strip_bytes_per_canvas": xstats.get("p2p_bytes_total", 0) / ncan,
                           "p2p_messages": xstats.get("p2p_messages_total", 0),
                           "final_gather_bytes_per_rank": xstats.get("final_gather_bytes_per_rank", 0),
                           "blocking_syncs_per_job": xstats.get("blocking_collectives", 0),
                           "gather": args.grid_gather,
                           "waves": xstats.get("waves"),
                           "per_rank": per_rank,
                           "per_rank_note": "tasks, generalised waves without a task (idle slots), seconds the rank's stream "
                                            "stood in receive waits (HIP events around each wait; host seconds under gloo), "
                                            "bytes sent and batches posted - of the last grid of the timed region",
                           "at_8_ranks": {"p2p_strip_bytes_per_canvas": plan8[8].p2p_bytes(0.25) / ncan,
                                          "p2p_messages": plan8[8].p2p_messages(),
                                          "whole_patch_allgather_bytes_per_canvas_round3": sum(
                                              4 * 3 * G.PATCH_SIZES[t[0]] ** 2 * 7 for w in plan8[8].waves for t in w) / ncan,
                                          "blocking_syncs_round3": len(plan8[8].waves)}},
                       "pipeline_steps": len(D.stage_waves([pos] * ncan, orient, (1, 2, 3), not args.no_pipeline)),
                       "parallelism": f"{world} rank(s): anti-diagonal waves of the three stages pipelined (a patch's "
                                      "stage s starts once its stage s-1 and its neighbours' stage s are done), dealt "
                                      "heaviest first with column affinity; finished patches stay on their rank, only the "
                                      "overlap strips travel (async point-to-point bundles per wave, waited for by the "
                                      "consuming task), one collective after the last wave"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "achieved_direct_equiv": direct_equiv, "traffic": None,
                         "kernel": f"whole patch pipeline per GPU: {issued_patch / 1e12:.2f} TFLOP issued on the matrix cores per "
                                   f"patch ({flop_patch / 1e12:.2f} of direct-convolution work), over the wall clock (includes "
                                   "host-side inpaint-tensor assembly, all-gathers and idle wave slots)"},
        })
    return results if rank == 0 else None


OTHER_CONFIGS = [   # (name, what, Unet kwargs, batch, image size): the UNets of the other BASELINE configs, at full dims
    ("configs[0]", "unconditional base UNet 64x64 at batch 1 (train_uncond.py:30-36; the sample_uncond.py:49-55 path)",
     dict(dim=256, dim_mults=(1, 2, 4, 8), cond_dim=512, num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
          layer_cross_attns=(F_, T_, T_, T_), lowres_cond=False, cond_on_text=False, text_embed_dim=None), 1, 64),
    ("configs[1]", "segmentation-conditioned base UNet 64x64 at batch 16 (train.py:30-41: 4 cond channels, text dim 3; "
                   "the sample_cond.py:40-48 path)",
     dict(dim=256, dim_mults=(1, 2, 3, 4), cond_dim=512, text_embed_dim=3, num_resnet_blocks=3, layer_attns=(F_, T_, T_, T_),
          layer_cross_attns=(F_, T_, T_, T_), cond_images_channels=4, lowres_cond=False, cond_on_text=True), 16, 64),
    ("configs[3] stage 3", "unet3 256->1024 at batch 8 (train_ultra_res.py:51-60): 96 % of a cascade's work",
     dict(**ULTRA_UNETS[3], lowres_cond=True, cond_on_text=False, text_embed_dim=None), 8, 1024),
]


def other_configs(device, lib, iters=2):
    """One UNet forward of the other BASELINE configurations at the reference's dims (random-init weights built in
    HBM): device ms by HIP events on the launch stream, the direct-convolution-equivalent and the issued TFLOP/s,
    launches per forward.  A forward, not a sampler step (no x0 / quantile / update: < 1 % of these steps)."""
    import gc

    import imagen_pytorch as ip
    from imagen_pytorch import _engine as E

    out = []
    for name, what, kw, B, S in OTHER_CONFIGS:
        with torch.device("meta"):
            u = ip.Unet(**kw)
        u = u.to_empty(device=device)
        gen = torch.Generator(device=device).manual_seed(7)
        with torch.no_grad():
            for p_ in u.parameters():
                p_.normal_(0, 0.02, generator=gen)
        with_text = bool(kw.get("cond_on_text"))
        h = u.engine(B, S, device, with_text=with_text)
        cc = kw.get("cond_images_channels", 0)
        x = torch.randn(B, 3, S, S, device=device, generator=gen)
        lr = torch.randn(B, 3, S, S, device=device, generator=gen) if kw["lowres_cond"] else None
        cond = torch.rand(B, cc, S, S, device=device, generator=gen) if cc else None
        t = torch.full((B,), 0.3, device=device)
        tl = torch.full((B,), -1.0, device=device) if kw["lowres_cond"] else None
        tok = hid = None
        if with_text:
            te = torch.tensor([0.0, 0.5, 0.2], device=device).reshape(1, 1, 3).repeat(B, 1, 1)   # sample_cond.py:37
            tok, hid = u.text_cond(h, te, None, False, device)
        y = torch.empty_like(x)

        def fwd():
            E.check(lib.kd_unet_forward(h, E.ptr(x), E.ptr(lr), E.ptr(cond), E.ptr(t), E.ptr(tl), E.ptr(tok), E.ptr(hid),
                                        E.ptr(y), E.current_stream()))

        fwd()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(iters):
            fwd()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / iters
        assert torch.isfinite(y).all(), name
        macs, issued, bf = lib.kd_unet_macs(h), lib.kd_unet_mfma_macs(h), lib.kd_unet_mfma_bf16_macs(h)
        out.append({"config": name, "what": what, "batch": B, "image_size": S, "forward_ms": ms,
                    "gmac_per_sample": macs / 1e9 / B, "direct_equiv_tflops": 2.0 * macs / ms / 1e9,
                    "issued_fp32_mfma_tflops": 2.0 * issued / ms / 1e9, "issued_bf16_mfma_tflops": 2.0 * bf / ms / 1e9,
                    "launches": lib.kd_unet_num_launches(h), "plan_hbm_gb": lib.kd_unet_hbm_bytes(h) / 1e9})
        del h, u, x, lr, cond, y
        gc.collect()
        torch.cuda.empty_cache()
    return out


def rccl_ranks(distributed, backend, device):
    """Number of ranks the collective library really connected: an all-reduce (sum) of ones."""
    if not distributed:
        return 1
    import torch.distributed as dist

    one = torch.ones(1, device=device if backend == "nccl" else "cpu", dtype=torch.int32)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through
    torch.distributed.run as a CHILD process - before this process has made any GPU call - relay its output
    (rank 0 prints the one JSON line) and exit with its code.  The reference does the same with
    torch.multiprocessing (sample_ultra_res.py:235-240, 493)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-kernel-classes", action="store_true",
                    help="skip the per-launch profile after the timed region (rocprofv3 runs: keeps only the steps in the trace)")
    ap.add_argument("--workload", choices=("sr", "grid"), default="sr",
                    help="sr: the headline metric (default); grid: ultra-res patch grid, patches/s")
    ap.add_argument("--grid-n", type=int, default=8)
    ap.add_argument("--canvases", type=int, default=1)
    ap.add_argument("--grid-steps", type=int, default=8, help="timesteps per stage for --workload grid")
    ap.add_argument("--grid-resample", type=int, default=1, help="inpaint_resample_times for --workload grid")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="--workload grid: keep the reference's barrier between the stages")
    ap.add_argument("--no-overlap", action="store_true",
                    help="grid: run the stage groups of a wave one after the other (default: the light stage-1/2 groups "
                         "on a side stream beside stage 3)")
    ap.add_argument("--grid-gather", choices=("all", "root"), default="all",
                    help="grid: the one collective after the last wave - all-gather of the final patches to every rank "
                         "(default; BASELINE's 'RCCL all-gather ... to reassemble the stitched canvas') or a gather to rank 0")
    ap.add_argument("--grid-batch", type=int, default=4,
                    help="--workload grid: same-wave patches of a rank per sample() call in stages 1-2 (1 = the reference's way, one "
                         "patch per call; 1 GPU, 8 timesteps: 2.48 patches/s at 1, 2.63 at 4, 2.62 at 8)")
    ap.add_argument("--no-cond-table", action="store_true",
                    help="compute the time conditioning in every step instead of restoring it from the per-schedule table "
                         "(profiles: keeps the one-off table build, 250 x 21 launches, out of a 7-step trace)")
    ap.add_argument("--fp32-mfma-gemms", action="store_true",
                    help="sr: the F(4x4,3x3) position GEMMs on the fp32 MFMA pipe (conv_buf_kernel) instead of the bf16x3 "
                         "kernel - the A/B of profiles/README.md")
    ap.add_argument("--v-form", type=int, default=0, choices=(0, 1, 2),
                    help="sr: Unet.gemm_bf16x3 (0 = the plan's rule per layer, 1 = V of every F(4x4,3x3) layer as bf16 planes, "
                         "2 = as fp32 split by the GEMM's loader waves)")
    ap.add_argument("--wino43-min-cin", type=int, default=0,
                    help="sr: Unet.wino43_min_cin (0 = the plan's default rule; 512 = the rule of the fp32 MFMA GEMMs)")
    ap.add_argument("--x3-linear", type=int, default=0,
                    help="sr: Unet.x3_linear (0 = the plan's default: token GEMMs / 1x1 convs with K >= 512 on the bf16x3 kernel; "
                         "-1 = on conv_buf_kernel, fp32 MFMA; n = K >= n)")
    ap.add_argument("--no-line-grid", action="store_true",
                    help="sr: leave the nested `grid` object (8x8 grid patches/s for 1 and 3 canvases) out of the line")
    ap.add_argument("--line-grid-steps", type=int, default=8, help="timesteps per stage of the nested grid runs")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="sr: leave `other_configs` (one forward of configs[0], [1] and unet3 at batch 8) out of the line")
    args = ap.parse_args()
    grid = args.workload == "grid"
    if args.steps is None:
        args.steps = 1 if grid else 20
    if args.warmup is None:
        args.warmup = 1 if grid else 3

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    distributed = world > 1
    # one process per GPU; KD_BENCH_BACKEND=gloo lets several ranks share one card for a rehearsal
    backend = os.environ.get("KD_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank)
    torch.cuda.set_device(device)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL; used for the barrier / max-reduce only
        else:
            dist.init_process_group(backend)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    nranks = rccl_ranks(distributed, backend, device)
    assert nranks == world, f"the collective backend connected {nranks} ranks, WORLD_SIZE is {world}"
    coll = "none (single process)" if not distributed else ("nccl = RCCL" if backend == "nccl" else backend)
    if grid:
        res = grid_workload(args, world, rank, device, distributed, barrier, [args.canvases], args.grid_steps, args.steps,
                            args.warmup)
        if rank == 0:
            res[0].update(rccl_ranks=nranks, collective_backend=coll)
            print(json.dumps(res[0]), flush=True)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    from imagen_pytorch import _engine as E
    from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes, log_snr_to_alpha_sigma, \
        beta_linear_log_snr

    lib = E.load()
    unet = build_unet(0)
    if args.fp32_mfma_gemms:
        unet.gemm_bf16x3 = -1
    elif args.v_form:
        unet.gemm_bf16x3 = args.v_form
    unet.wino43_min_cin = args.wino43_min_cin
    unet.x3_linear = args.x3_linear
    handle = unet.engine(BATCH, SIZE, device, with_text=False)
    macs = lib.kd_unet_macs(handle)            # algorithmic: the direct convolutions the reference computes
    mfma_macs = lib.kd_unet_mfma_macs(handle)  # issued on the matrix cores (Winograd layers: 16/36 of theirs)
    mfma_bf16_macs = lib.kd_unet_mfma_bf16_macs(handle)  # bf16 MACs of the bf16x3 GEMMs (6 per fp32 MAC; not in mfma_macs)
    flop_per_step = 2.0 * macs
    launches = lib.kd_unet_num_launches(handle)
    cond_launches = lib.kd_unet_num_cond_launches(handle)

    x, lowres, lowres_noise, cond = synthetic_inputs(BATCH, device, seed=1234 + rank)
    ls_lr = beta_linear_log_snr(torch.full((BATCH,), 0.2))
    a, s = log_snr_to_alpha_sigma(ls_lr)
    lowres = (a.to(device)[:, None, None, None] * lowres + s.to(device)[:, None, None, None] * lowres_noise).contiguous()
    lowres_log_snr = ls_lr.to(device)
    sched = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=T_SCHED)  # stage 2: cosine
    tables = sched.step_tables()
    sc = E.kd_schedule_t()
    sc.T = T_SCHED
    for name, v in tables.items():
        setattr(sc, name, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))
    sa = E.kd_sample_args_t()
    sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times = 0, 1, 0.95, 1
    sa.d_lowres, sa.d_lowres_log_snr, sa.d_cond_images = E.ptr(lowres), E.ptr(lowres_log_snr), E.ptr(cond)
    sa.lowres_log_snr_uniform, sa.lowres_log_snr_value = 1, float(ls_lr[0])   # one augmentation level (0.2) for the batch
    sa.cond_table = -1 if args.no_cond_table else 0
    sa.seed = 1234 + rank  # per-step noise: on-device Philox inside the fused DDPM-update kernel
    sa.use_graph = 0 if args.no_graph else 1

    def run_steps(k0, n):
        """n consecutive denoising steps starting at schedule index k0 (wraps around T)."""
        while n > 0:
            k0 %= T_SCHED
            m = min(n, T_SCHED - k0)
            E.check(lib.kd_sample_steps(handle, C.byref(sc), C.byref(sa), E.ptr(x), k0, k0 + m, E.current_stream()))
            k0 += m
            n -= m

    run_steps(0, args.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(args.warmup, args.steps)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the graph is launched on
    assert torch.isfinite(x).all(), "sampler state went non-finite"
    if distributed:
        tmax = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # what the conditioning table costs a cold 250-step sample(): all T rows rebuilt, device time by HIP events
    cond_build = None
    if not args.no_cond_table:
        built, rows, runs = C.c_int(0), C.c_int(0), C.c_int(0)
        E.check(lib.kd_sample_build_cond_table(handle, C.byref(sc), C.byref(sa), 0, T_SCHED, 1, C.byref(built), E.current_stream()))
        ms = float(lib.kd_unet_cond_table_build_ms(handle, C.byref(rows), C.byref(runs)))
        cond_build = {"ms": ms, "rows": int(built.value), "schedule_steps": T_SCHED,
                      "runs_of_the_conditioning_ops": int(runs.value),
                      # > 0: the table was refused (size cap = 1/8 of the free HBM, or the allocation failed) and the
                      # timed steps computed their conditioning themselves
                      "refused_bytes": int(lib.kd_unet_cond_table_refused_bytes(handle))}

    # the >= 6x target of BASELINE.json is patch throughput of the 8x8 grid: measured by the same command, after the
    # timed region of the headline metric, on every rank (collective inside); 1 canvas (dependency bound 4.27x at 8
    # ranks) and 3 canvases in flight (bound 6.2x)
    grid_res = None
    if not args.no_line_grid:
        grid_res = grid_workload(args, world, rank, device, distributed, barrier, [1, 3], args.line_grid_steps, 1, 1)

    others = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        others = other_configs(device, lib)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        dev_ms_per_step = dev_ms / args.steps
        step_issued = 2.0 * mfma_macs / (dev_ms_per_step * 1e-3) / 1e12
        step_issued_bf16 = 2.0 * mfma_bf16_macs / (dev_ms_per_step * 1e-3) / 1e12
        # share of the step's time the matrix pipe would be busy at its nominal rates: fp32 MFMA work + bf16 MFMA work
        step_pipe_frac = step_issued / FP32_PEAK_TFLOPS + step_issued_bf16 / BF16_PEAK_TFLOPS
        step_direct = flop_per_step / (dev_ms_per_step * 1e-3) / 1e12
        kernels = kernel_classes(lib, handle) if world == 1 and not args.no_kernel_classes else None
        # the dominant kernel: whichever of the ResnetBlock 3x3-conv kernels takes most of the step in this plan (since the end
        # of round 5 the bf16x3 position GEMMs of F(4x4,3x3); the fused F(2x2,3x3) kernel where the plan keeps layers on it)
        dom = max((k for k in (kernels or []) if k["kernel"] in (DOMINANT, X3_ALL)), key=lambda k: k["ms"], default=None)
        # committed profile artefacts of the same command (profiles/): rocprofv3's average launch duration of the
        # dominant kernel and the PMC traffic.  They are NOT measured in this run and are labelled as such.
        prof = {}
        pf = ROOT / "profiles" / "current.json"
        if pf.exists():
            prof = json.loads(pf.read_text())
        roof = {"bound": "mfma", "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s"}
        if dom:
            n = dom["launches"]
            x3 = dom["kernel"] == X3_ALL
            same = ("gemm_bf16x3" in prof.get("dominant_kernel", "")) == x3   # the profile set is of this plan's dominant kernel
            if x3:
                work = (f"achieved = FLOPs the kernel ISSUES on the bf16 MFMA pipe per launch ({dom['issued_tflop_per_step'] / n * 1e3:.1f} "
                        f"GFLOP average over its {n} launches per step = 2 x 6 bf16 products x rows x Cout x K: six bf16 MACs per "
                        "fp32 MAC; rows x K = 36 Winograd positions x tiles x Cin for the 56 position GEMMs of the F(4x4,3x3) layers, "
                        "36/144 of the direct 3x3 convolution's MACs, and pixels x Cin for the token GEMMs / 1x1 convs) / average "
                        "launch duration measured live with HIP events around each launch (kd_unet_profile, on the launch stream; "
                        "the sum_slabs_kernel launches that add the k-parts of left-over tiles are another kernel and not in it: "
                        "sum_launches_ms); peak = the dense bf16 MFMA rate at 2.4 GHz - under this load the chip sustains "
                        "1.9-2.3 GHz, a bare six-product loop issues 1.24-1.37 PFLOP/s (profiles/README.md) - and the launches with "
                        "Cout = 128 on the 128 x 128 and 256 x 256 maps wait for HBM, not for the pipe (V + D = 2.4 GB per launch at "
                        "256 x 256: traffic, against algorithmic_bytes_per_launch = V + D + weights / A + Y + weights; by_side prices each group of "
                        "launches against both roofs: mfma_frac of 2516.6 TFLOP/s and hbm_frac of 8 TB/s over the same time); achieved_fp32_equiv = the same work as fp32 TFLOP/s (a sixth), achieved_direct_equiv "
                        "prices the time against the direct-convolution FLOPs of SURVEY §8d - neither is a utilisation")
                roof.update(peak=BF16_PEAK_TFLOPS, peak_dtype="bf16 (dense MFMA)", achieved_fp32_equiv=dom["achieved_fp32_equiv"])
            else:
                work = (f"achieved = FLOPs the kernel ISSUES on the fp32 MFMA pipe per launch ({dom['issued_tflop_per_step'] / n * 1e3:.1f} "
                        f"GFLOP average over its {n} launches per step = 2 x 16 Winograd positions x tiles x Cout x Cin, i.e. "
                        "16/36 of the direct 3x3 convolution the reference computes) / average launch duration measured "
                        "live with HIP events around each launch (kd_unet_profile, on the launch stream); "
                        "achieved_direct_equiv prices the same time against the direct-convolution FLOPs of SURVEY §8d "
                        "and can exceed the peak - it is not a utilisation")
            roof.update(
                kernel=dom["kernel"],
                achieved=dom["achieved"], frac=dom["frac"],
                achieved_direct_equiv=dom["achieved_direct_equiv"],
                launches_per_step=n, avg_launch_us=dom["avg_us"], share_of_step=dom["share"],
                sum_launches_ms=dom.get("sum_launches_ms"),
                algorithmic_bytes_per_launch=dom.get("algorithmic_bytes_per_launch"), by_side=dom.get("by_side"),
                work=work,
                rocprof_avg_launch_us=prof.get("dominant_avg_us") if same else None,
                traffic=prof.get("dominant_bytes_per_launch") if same else None,
                traffic_source=("profile-derived, not measured in this run: " + prof["source"]) if prof.get("source") and same else None)
        else:  # multi-GPU runs / --no-kernel-classes: no per-launch profile, whole-step pipe utilisation instead
            # (nearly all matrix work of the plan is bf16 MFMA now: the step's bf16 rate against the bf16 peak, per GPU; the
            # fp32-pipe share and the combined busy fraction are in roofline.step)
            if step_issued_bf16 / BF16_PEAK_TFLOPS >= step_issued / FP32_PEAK_TFLOPS:
                roof.update(kernel="whole denoising step, per GPU (per-launch profile skipped)", peak=BF16_PEAK_TFLOPS,
                            peak_dtype="bf16 (dense MFMA)", achieved=step_issued_bf16, frac=step_issued_bf16 / BF16_PEAK_TFLOPS,
                            achieved_direct_equiv=step_direct, traffic=None)
            else:   # (a plan on the fp32 pipe: --fp32-mfma-gemms)
                roof.update(kernel="whole denoising step, per GPU (per-launch profile skipped)", achieved=step_issued,
                            frac=step_issued / FP32_PEAK_TFLOPS, achieved_direct_equiv=step_direct, traffic=None)
        roof["step"] = {
            "issued_tflops": step_issued, "issued_bf16_tflops": step_issued_bf16, "frac_issued": step_pipe_frac,
            "direct_equiv_tflops": step_direct, "device_ms": dev_ms_per_step,
            "algorithmic_tflop_per_step": flop_per_step / 1e12, "issued_tflop_per_step": 2.0 * mfma_macs / 1e12,
            "issued_bf16_tflop_per_step": 2.0 * mfma_bf16_macs / 1e12,
            "traffic_bytes_per_step": prof.get("bytes_per_step"),
            "note": "one denoising-step graph = UNet forward + x0 / quantile / DDPM update; device time by HIP events "
                    "on the launch stream; issued = what the conv / GEMM launches put on the fp32 matrix pipe "
                    "(kd_unet_mfma_macs), issued_bf16 = the bf16 MFMA work of the bf16x3 GEMMs (kd_unet_mfma_bf16_macs, six "
                    "bf16 MACs per fp32 MAC), frac_issued = issued / 157.3 + issued_bf16 / 2516.6 (the share of the step the "
                    "matrix pipe is busy at nominal rates), direct_equiv = 2 x 229.2 GMAC/sample x 16 of SURVEY §8d / time"}
        roof["kernels"] = kernels
        out = {
            "metric": "denoising-steps/sec (64->256 SR UNet, bs16)",
            "value": world * args.steps / elapsed,
            "unit": "denoising-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: stage-2 super-res UNet 64->256 (train_ultra_res.py:39-48, "
                                   "3 cond channels), batch 16 per GPU, cosine schedule T=250, dynamic thresholding, "
                                   "random-init weights, Philox noise on device, hipGraph-replayed step",
                       "arithmetic": ("fp32 values and fp32 accumulation everywhere"
                                      + ("; the position GEMMs of the F(4x4,3x3) layers form each fp32 product from six exact "
                                         "bf16 MFMA products of three-piece operands (a = ah + am + al exactly) - error against "
                                         "fp64 at or below the fp32 MFMA path's (tests/test_kernels_gpu.py, "
                                         "test_fullsize_gpu.py), same parity bounds" if mfma_bf16_macs else "")),
                       "batch_per_gpu": BATCH, "image_size": SIZE, "launches_per_step": launches,
                       "launches_per_step_from_cond_table": launches if args.no_cond_table else launches - cond_launches + 1,
                       "cond_table": ("off" if args.no_cond_table else
                                      f"{cond_launches} conditioning launches (time MLPs, FiLM scale / shift GEMM, tokens, cross-attention K / V) "
                                      "replaced by one gather per step; the per-launch profile in roofline.kernels runs all of them"),
                       "cond_table_build_ms": cond_build,
                       "parallelism": f"{world} independent batch replicas (no data-path collective)"},
            "roofline": roof,
            "rccl_ranks": nranks, "collective_backend": coll,
        }
        if grid_res:
            out["grid"] = {
                "what": "BASELINE configs[4] measured by this same command after the headline region: patches/s of the 8x8 "
                        "grid of 1024-px patches (3-stage cascade, strong scaling: the same patches whatever N is), for 1 canvas "
                        "and for 3 canvases in flight; compare `patches_per_s` across the N = 1/2/4/8 lines",
                "timesteps_per_stage": args.line_grid_steps,
                **{f"canvases_{r['config']['canvases']}": {
                    "patches_per_s": r["value"], "patches": r["config"]["patches"], "seconds": r["ms_per_step"] / 1e3,
                    "schedule_slots": r["config"]["schedule_slots"], "pipeline_steps": r["config"]["pipeline_steps"],
                    "schedule_speedup_bound": r["config"]["schedule_speedup_bound"],
                    "exchange": r["config"]["exchange"],
                    "mfma_frac_per_gpu": r["roofline"]["frac"]} for r in grid_res}}
        if others:
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            cpu, first = cpu_baseline(unet)
            out["cpu_baseline"] = cpu
            out.update(engine_parity(unet, first, device, lib))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
