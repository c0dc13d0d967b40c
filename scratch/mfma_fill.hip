// Micro-benchmark: how many VALU / LDS fillers hide in the shadow of v_mfma_f32_32x32x2_f32 (fp32 MFMA,
// 64 cycles per SIMD) at one and two waves per SIMD?  Prints shader cycles per MFMA (s_memtime) for a
// loop of 16 MFMAs on 4 accumulators with NF fillers of one kind after every MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_fill scratch/mfma_fill.hip && scratch/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_add_f32   1: v_exp_f32   2: ds_write_b32   3: ds_read_b32 (no wait)   4: v_fma_f32 dependent chain
template <int NF, int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void fill_kernel(float* out, unsigned long long* cyc, unsigned long long* span,
                                                          int iters) {
  __shared__ float lds[WAVES * 64 * 4];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  float* lp = lds + threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(b));
        if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 7]));
        if constexpr (KIND == 2) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)lp), "v"(f[k & 7]) : "memory");
        if constexpr (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(f[k & 7]) : "v"((unsigned)(size_t)lp) : "memory");
        if constexpr (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[0]) : "v"(b));
      }
    }
    if constexpr (KIND == 2 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  // end-to-end: from the first wave's start to the last wave's end of this workgroup (all waves of the CU)
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&span[blockIdx.x * 2], t0);
    atomicMax(&span[blockIdx.x * 2 + 1], t1);
  }
}

template <int NF, int KIND, int WAVES>
static void run(const char* name) {
  const int blocks = 256, iters = 2000;
  float* out;
  unsigned long long* cyc;
  hipMalloc((void**)&out, blocks * WAVES * 64 * 4);
  hipMalloc((void**)&cyc, blocks * 8);
  unsigned long long* span;
  hipMalloc((void**)&span, blocks * 16);
  unsigned long long hs[512];
  for (int i = 0; i < blocks; ++i) { hs[2 * i] = ~0ull; hs[2 * i + 1] = 0; }
  hipLaunchKernelGGL((fill_kernel<NF, KIND, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, span, iters);
  hipMemcpy(span, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL((fill_kernel<NF, KIND, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, span, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks; ++i) sum += (double)h[i];
  hipMemcpy(hs, span, sizeof(hs), hipMemcpyDeviceToHost);
  double sp = 0;
  for (int i = 0; i < blocks; ++i) sp += (double)(hs[2 * i + 1] - hs[2 * i]);
  printf("%-14s NF=%2d waves/SIMD=%d : %.1f cycles per MFMA (wave 0 alone), %.1f cycles of the SIMD per MFMA (all its waves)\n", name,
         NF, WAVES / 4, sum / blocks / iters / 16, sp / blocks / iters / 16 / (WAVES / 4));
  hipFree(out);
  hipFree(cyc);
  hipFree(span);
}

template <int KIND, int WAVES>
static void sweep(const char* name) {
  run<0, KIND, WAVES>(name);
  run<2, KIND, WAVES>(name);
  run<4, KIND, WAVES>(name);
  run<6, KIND, WAVES>(name);
  run<8, KIND, WAVES>(name);
  run<12, KIND, WAVES>(name);
}

int main() {
  sweep<0, 4>("v_add_f32");
  sweep<1, 4>("v_exp_f32");
  sweep<2, 4>("ds_write_b32");
  sweep<3, 4>("ds_read_b32");
  sweep<4, 4>("v_fma dep");
  sweep<0, 8>("v_add_f32");
  sweep<1, 8>("v_exp_f32");
  sweep<2, 8>("ds_write_b32");
  return 0;
}
