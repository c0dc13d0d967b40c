"""Builds the conditioning table of the headline plan (batch 16, T = 250) and of a batch-2 plan: device time, rows, runs."""
import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'kidney-diffusion_amd')
import bench
from imagen_pytorch import _engine as E
from imagen_pytorch.imagen_pytorch import GaussianDiffusionContinuousTimes
lib = E.load(); dev = torch.device('cuda:0')
for B in (16, 2):
    u = bench.build_unet(0)
    h = u.engine(B, 256, dev, with_text=False)
    sched = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=250)
    tables = sched.step_tables(); sc = E.kd_schedule_t(); sc.T = 250
    for name, v in tables.items(): setattr(sc, name, v.numpy().ctypes.data_as(C.POINTER(C.c_float)))
    sa = E.kd_sample_args_t(); sa.objective, sa.dynamic_threshold, sa.percentile, sa.resample_times = 0, 1, 0.95, 1
    ll = torch.full((B,), -1.0, device=dev); sa.d_lowres_log_snr = E.ptr(ll); sa.lowres_log_snr_uniform, sa.lowres_log_snr_value = 1, -1.0
    built, rows, runs = C.c_int(0), C.c_int(0), C.c_int(0)
    for rep in range(2):
        E.check(lib.kd_sample_build_cond_table(h, C.byref(sc), C.byref(sa), 0, 250, 1, C.byref(built), E.current_stream()))
        ms = lib.kd_unet_cond_table_build_ms(h, C.byref(rows), C.byref(runs))
        print(f"batch {B}: build {ms:.2f} ms, rows {built.value}, runs {runs.value}, cond launches {lib.kd_unet_num_cond_launches(h)}")
