# Experiment: the batch-16 denoising step as two batch-8 plans on two streams (HBM-bound kernels of one half
# under the MFMA-bound kernels of the other) against one batch-16 plan.
import sys, time, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'kidney-diffusion_amd')
import bench
from imagen_pytorch import _engine as E
from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes, log_snr_to_alpha_sigma, beta_linear_log_snr

lib = E.load(); dev = torch.device('cuda:0')
unet = bench.build_unet(0)
sched = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=250)
tables = sched.step_tables()
sc = E.kd_schedule_t(); sc.T = 250
for name, v in tables.items():
    setattr(sc, name, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))

def make(B, replica, seed):
    h = unet.engine(B, 256, dev, with_text=False, replica=replica)
    x, lowres, ln, cond = bench.synthetic_inputs(B, dev, seed=seed)
    ls = beta_linear_log_snr(torch.full((B,), 0.2)); a, s = log_snr_to_alpha_sigma(ls)
    lowres = (a.to(dev)[:, None, None, None] * lowres + s.to(dev)[:, None, None, None] * ln).contiguous()
    sa = E.kd_sample_args_t()
    sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times = 0, 1, 0.95, 1
    lls = ls.to(dev)
    sa.d_lowres, sa.d_lowres_log_snr, sa.d_cond_images = E.ptr(lowres), E.ptr(lls), E.ptr(cond)
    sa.seed = seed; sa.use_graph = 1
    return dict(h=h, x=x, sa=sa, keep=(lowres, lls, cond))

def run(parts, streams, k0, n):
    for p, st in zip(parts, streams):
        with torch.cuda.stream(st):
            E.check(lib.kd_sample_steps(p['h'], C.byref(sc), C.byref(p['sa']), E.ptr(p['x']), k0, k0 + n, E.current_stream()))

for label, parts in (("one plan, batch 16", [make(16, 0, 1)]), ("two plans, batch 8 each, two streams", [make(8, 0, 2), make(8, 1, 3)])):
    streams = [torch.cuda.Stream() for _ in parts]
    run(parts, streams, 0, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(parts, streams, 3, 20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {dt / 20 * 1e3:.2f} ms per batch-16 step, {20 / dt:.2f} steps/s")
