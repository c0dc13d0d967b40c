"""Driver of mfma_shape_probe.hip: time of the six-product inner loop on 32x32x16 and on 16x16x32 bf16 MFMAs, one workgroup of
8 waves per CU, sustained for ~0.1-0.2 s each (the clock settles), alternating three times."""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmfma_probe.so"))
lib.probe_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(0).multi_processor_count
n = 6 * 3 * 64 * 16
# random bf16 values of magnitude ~1 (random mantissas and signs, exponents 120..130)
g = torch.Generator(device=dev).manual_seed(1)
vals = (torch.randn(2 * n, device=dev, generator=g)).to(torch.bfloat16).view(torch.int16).view(torch.int32)[:n].contiguous()
out = torch.empty(cus * 512, device=dev)
s = torch.cuda.current_stream().cuda_stream
iters = 20000
flop_per_iter = 8 * 6 * 2.0 * 64 * 64 * 32   # per workgroup: 8 waves x six products of a 64 x 64 x 32 block
for rep in range(3):
    for which in (32, 16):
        lib.probe_run(which, vals.data_ptr(), out.data_ptr(), cus, 200, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            lib.probe_run(which, vals.data_ptr(), out.data_ptr(), cus, iters, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        print(f"rep {rep} mfma {'32x32x16' if which == 32 else '16x16x32'}: {ms:8.3f} ms  {cus * iters * flop_per_iter / ms / 1e9:8.1f} TFLOP/s bf16"
              f"  = {cus * iters * flop_per_iter / 6 / ms / 1e9:6.1f} fp32-equivalent", flush=True)
