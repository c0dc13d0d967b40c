"""EXPERIMENT driver: the bf16x3 GEMM of gemm_bf16x3.hip against torch.bmm (fp32 MFMA through rocBLAS/hipBLASLt) and an
fp64 product, on the shapes of the engine's batched F(4x4,3x3) GEMMs (G = 36 positions, M = tiles of the batch,
N = Cout, K = Cin).  Prints one line per shape / variant: time, fp32-equivalent TFLOP/s (2 G M N K / t), max and rms
error relative to the rms of the exact result.

  python scratch/bf16x3/bf16x3_bench.py            (needs libbf16x3.so beside it; built by the command in the .hip header)
"""
import ctypes
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libbf16x3.so"))
lib.bf16x3_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.bf16x3_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    shapes = [(36, 4096, 512, 512), (36, 4096, 512, 1024), (36, 1024, 1024, 1024), (36, 1024, 1024, 2048),
              (1, 8192, 8192, 8192)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for G, M, N, K in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.randn(G, M, K, device=dev, generator=g)
        B = torch.randn(G, N, K, device=dev, generator=g) * 0.05
        A3 = torch.empty(3, G, K // 16, M, 16, device=dev, dtype=torch.int16)   # k-chunk-major planes
        B3 = torch.empty(3, G, K // 16, N, 16, device=dev, dtype=torch.int16)
        C = torch.empty(G, M, N, device=dev)
        assert lib.bf16x3_split(A.data_ptr(), A3.data_ptr(), G, M, K, stream) == 0
        assert lib.bf16x3_split(B.data_ptr(), B3.data_ptr(), G, N, K, stream) == 0
        torch.cuda.synchronize()
        # the three planes add back to the fp32 value exactly (or to within the last bit of a denormal tail)
        rec = sum(p.view(torch.bfloat16).double() for p in A3).permute(0, 2, 1, 3).reshape(G, M, K)
        print(f"G{G} M{M} N{N} K{K}: split residual {float((rec - A.double()).abs().max()):.2e}", flush=True)
        del rec
        # exact reference on a slice (fp64 on the whole thing is slow and large)
        gs = 0
        ms = min(M, 512)
        exact = A[gs, :ms].double() @ B[gs].double().T
        scale = float(exact.pow(2).mean().sqrt())
        flops = 2.0 * G * M * N * K

        def report(name, ms_t, out):
            err = (out[gs, :ms].double() - exact)
            print(f"  {name:14s} {ms_t:8.3f} ms  {flops / ms_t / 1e9:7.1f} TF  max {float(err.abs().max()) / scale:.2e}"
                  f"  rms {float(err.pow(2).mean().sqrt()) / scale:.2e}", flush=True)

        Bt = B.transpose(1, 2)
        t = timed(lambda: torch.bmm(A, Bt, out=C))
        report("torch.bmm fp32", t, C)
        for nprod in (6, 3, 1, 16, 13, 11):   # 1x: four loader waves issue the DMAs
            C.zero_()
            rc = lib.bf16x3_gemm(A3.data_ptr(), B3.data_ptr(), C.data_ptr(), G, M, N, K, nprod, stream)
            assert rc == 0, rc
            t = timed(lambda: lib.bf16x3_gemm(A3.data_ptr(), B3.data_ptr(), C.data_ptr(), G, M, N, K, nprod, stream))
            report(f"bf16x3 n={nprod % 10}" + (" +loaders" if nprod > 10 else ""), t, C)
        t = timed(lambda: lib.bf16x3_split(A.data_ptr(), A3.data_ptr(), G, M, K, stream))
        print(f"  split of A     {t:8.3f} ms  ({A.numel() * 10 / t / 1e6:.0f} GB/s)", flush=True)
        del A, B, A3, B3, C


if __name__ == "__main__":
    main()
