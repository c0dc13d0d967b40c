# Headline step time with the conditioning table on / off in ONE process (same box, same plan)
import sys, time, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'kidney-diffusion_amd')
import bench
from imagen_pytorch import _engine as E
from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes, log_snr_to_alpha_sigma, beta_linear_log_snr
lib = E.load(); dev = torch.device('cuda:0'); B, S, T = 16, 256, 250
unet = bench.build_unet(0); h = unet.engine(B, S, dev, with_text=False)
x, lowres, ln, cond = bench.synthetic_inputs(B, dev)
ls = beta_linear_log_snr(torch.full((B,), 0.2)); lsd = ls.to(dev)
tables = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=T).step_tables()
sc = E.kd_schedule_t(); sc.T = T
for n, v in tables.items(): setattr(sc, n, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))
def run(table, steps=40):
    sa = E.kd_sample_args_t()
    sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times = 0, 1, 0.95, 1
    sa.d_lowres, sa.d_lowres_log_snr, sa.d_cond_images = E.ptr(lowres), E.ptr(lsd), E.ptr(cond)
    sa.lowres_log_snr_uniform, sa.lowres_log_snr_value, sa.cond_table = 1, float(ls[0]), table
    sa.seed, sa.use_graph = 5, 1
    E.check(lib.kd_sample_steps(h, C.byref(sc), C.byref(sa), E.ptr(x), 0, 5, E.current_stream()))
    torch.cuda.synchronize(); t0 = time.time()
    E.check(lib.kd_sample_steps(h, C.byref(sc), C.byref(sa), E.ptr(x), 5, 5 + steps, E.current_stream()))
    torch.cuda.synchronize(); return (time.time() - t0) * 1e3 / steps
for t in (0, -1, 0, -1, 0, -1):
    print("cond_table", t, f"{run(t):.3f} ms/step")
