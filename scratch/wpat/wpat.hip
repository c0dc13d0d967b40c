// Write-pattern microbenchmark for the F(4x4,3x3) input transform's plane-form output (round 5):
// is wino4_in_kernel<true> bound by WHERE its stores go?  Same thread -> address map as the kernel, no arithmetic.
//   mode 0: the kernel's layout  [plane][pos][chunk][row][16]           (a wave: 108 pieces of 256 B, 4 MB apart)
//   mode 1: tile-major layout    [mt][chunk][pos][plane][256 rows][16]  (a wave: 108 pieces of 256 B inside 864 KB)
//   mode 2: linear streaming stores of the same byte count (dword per lane)
//   mode 3: as 2 with 16-byte stores
// usage: wpat <nt> <C> [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// OCC4: at most four waves per SIMD (what wino4_in_kernel<true>'s 104 VGPRs allow); VALU: ~the kernel's vector work per thread
// (72 SiLUs through exp2 / rcp, ~700 plain operations), results folded into the stored value
template <int MODE, bool READ, bool VALU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void wk4(uint32_t* __restrict__ V, const float* __restrict__ x, int64_t nt, int C) {
  const int C2 = C >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nt * C2) return;
  const int64_t w = idx >> 6;
  const int lane = (int)(idx & 63), nkc = C / 16;
  const int c = (int)(w % nkc) * 16 + (lane & 7) * 2;
  const int64_t t = (w / nkc) * 8 + (lane >> 3);
  float2 d[36];
  {
    const int Wt = 64, Ht = 64;
    const int tx = (int)(t % Wt), ty = (int)((t / Wt) % Ht);
    const int64_t b = t / (Wt * Ht);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        int iy = 4 * ty - 1 + r, ix = 4 * tx - 1 + s;
        iy = iy < 0 ? 0 : (iy > 4 * Ht - 1 ? 4 * Ht - 1 : iy);
        ix = ix < 0 ? 0 : (ix > 4 * Wt - 1 ? 4 * Wt - 1 : ix);
        d[r * 6 + s] = *(const float2*)(x + (((b * 4 * Ht + iy) * 4 * Wt + ix) * (int64_t)C + c));
      }
  }
  if (VALU) {
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      float a = d[i].x * 1.01f + 0.5f, bq = d[i].y * 0.99f - 0.5f;
      a = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * a));
      bq = bq * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * bq));
      d[i] = make_float2(a, bq);
    }
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 30; ++i) {
        d[i].x = fmaf(d[i + 6].x, 4.0f, d[i].x) - 5.0f * d[i + 3].x;
        d[i].y = fmaf(d[i + 6].y, 4.0f, d[i].y) - 5.0f * d[i + 3].y;
      }
  }
  const int64_t pstride = nt * C2, plane = 36 * pstride;
  uint32_t* out = V + ((int64_t)(c / 16) * nt + t) * 8 + (c % 16) / 2;
#pragma unroll
  for (int p = 0; p < 36; ++p) {
    uint32_t* o = out + (int64_t)p * pstride;
    const uint32_t v0 = __float_as_uint(d[p].x), v1 = __float_as_uint(d[p].y);
    o[0] = v0;
    o[plane] = v1;
    o[2 * plane] = v0 ^ v1;
  }
}

template <int MODE, bool READ>
__global__ __launch_bounds__(256) void wk(uint32_t* __restrict__ V, const float* __restrict__ x, int64_t nt, int C) {
  const int C2 = C >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nt * C2) return;
  const int64_t w = idx >> 6;
  const int lane = (int)(idx & 63), nkc = C / 16;
  const int c = (int)(w % nkc) * 16 + (lane & 7) * 2;
  const int64_t t = (w / nkc) * 8 + (lane >> 3);
  uint32_t val = (uint32_t)idx;
  if (READ) {   // the 36 patch reads of the transform (8 bytes each), folded into the value
    // tile t of a (nt = B * 64 * 64 / 16 ...) map: use a synthetic 64 x 64 tile grid per image
    const int Wt = 64, Ht = 64;
    const int tx = (int)(t % Wt), ty = (int)((t / Wt) % Ht);
    const int64_t b = t / (Wt * Ht);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        int iy = 4 * ty - 1 + r, ix = 4 * tx - 1 + s;
        iy = iy < 0 ? 0 : (iy > 4 * Ht - 1 ? 4 * Ht - 1 : iy);
        ix = ix < 0 ? 0 : (ix > 4 * Wt - 1 ? 4 * Wt - 1 : ix);
        const float2 v = *(const float2*)(x + (((b * 4 * Ht + iy) * 4 * Wt + ix) * (int64_t)C + c));
        acc += v.x + v.y;
      }
    val = __float_as_uint(acc);
  }
  if (MODE == 0) {
    const int64_t pstride = nt * C2, plane = 36 * pstride;
    uint32_t* out = V + ((int64_t)(c / 16) * nt + t) * 8 + (c % 16) / 2;
#pragma unroll
    for (int p = 0; p < 36; ++p) {
      uint32_t* o = out + (int64_t)p * pstride;
      o[0] = val + p;
      o[plane] = val + p + 1;
      o[2 * plane] = val + p + 2;
    }
  } else if (MODE == 1) {
    // [mt][chunk][pos][plane][256][16] bf16 = dwords [..][256][8]
    const int64_t mt = t >> 8, row = t & 255;
    uint32_t* out = V + (((mt * nkc + c / 16) * 36) * 3) * (256 * 8) + row * 8 + (c % 16) / 2;
#pragma unroll
    for (int p = 0; p < 36; ++p) {
      uint32_t* o = out + (int64_t)p * 3 * (256 * 8);
      o[0] = val + p;
      o[256 * 8] = val + p + 1;
      o[2 * 256 * 8] = val + p + 2;
    }
  } else if (MODE == 2) {
    // each thread writes 108 dwords; wave-contiguous 256 B, consecutive instructions consecutive 256-B blocks
    uint32_t* out = V + (w * 108) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 108; ++p) out[p * 64] = val + p;
  } else {
    uint4* out = (uint4*)V + (w * 27) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 27; ++p) out[p * 64] = make_uint4(val, val + p, val + 1, val + 2);
  }
}

template <int MODE, bool READ>
float run(uint32_t* V, const float* x, int64_t nt, int C, int iters) {
  const int64_t total = nt * (C / 2);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((wk<MODE, READ>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, V, x, nt, C);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wk<MODE, READ>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, V, x, nt, C);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

template <bool VALU>
float run4(uint32_t* V, const float* x, int64_t nt, int C, int iters) {
  const int64_t total = nt * (C / 2);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((wk4<0, true, VALU>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, V, x, nt, C);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((wk4<0, true, VALU>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, V, x, nt, C);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t nt = argc > 1 ? atoll(argv[1]) : 4096 * 16;   // tiles (16 images of 64 x 64 tiles = 256 x 256 pixels... scaled)
  const int C = argc > 2 ? atoi(argv[2]) : 512;
  const int iters = argc > 3 ? atoi(argv[3]) : 10;
  const size_t vbytes = (size_t)nt * C * 36 * 6;
  uint32_t* V; float* x;
  CK(hipMalloc(&V, vbytes + 4096));
  const size_t xbytes = (size_t)(nt / 4096 + 1) * 256 * 256 * C * 4;
  CK(hipMalloc(&x, xbytes));
  CK(hipMemset(x, 0, xbytes));
  printf("nt %lld C %d: V %.1f MB, x %.1f MB\n", (long long)nt, C, vbytes / 1e6, (double)nt * 16 * C * 4 / 1e6);
  const char* names[4] = {"kernel layout [plane][pos][chunk][row][16]", "tile-major   [mt][chunk][pos][plane][256][16]", "linear dword stores", "linear 16-byte stores"};
  float t;
  t = run<0, false>(V, x, nt, C, iters); printf("  write only  %-48s %8.1f us  %6.2f TB/s\n", names[0], t, vbytes / t / 1e6);
  t = run<1, false>(V, x, nt, C, iters); printf("  write only  %-48s %8.1f us  %6.2f TB/s\n", names[1], t, vbytes / t / 1e6);
  t = run<2, false>(V, x, nt, C, iters); printf("  write only  %-48s %8.1f us  %6.2f TB/s\n", names[2], t, vbytes / t / 1e6);
  t = run<3, false>(V, x, nt, C, iters); printf("  write only  %-48s %8.1f us  %6.2f TB/s\n", names[3], t, vbytes / t / 1e6);
  t = run<0, true>(V, x, nt, C, iters);  printf("  read+write  %-48s %8.1f us  %6.2f TB/s (writes)\n", names[0], t, vbytes / t / 1e6);
  t = run<1, true>(V, x, nt, C, iters);  printf("  read+write  %-48s %8.1f us  %6.2f TB/s (writes)\n", names[1], t, vbytes / t / 1e6);
  t = run<2, true>(V, x, nt, C, iters);  printf("  read+write  %-48s %8.1f us  %6.2f TB/s (writes)\n", names[2], t, vbytes / t / 1e6);
  t = run4<false>(V, x, nt, C, iters);   printf("  read+write, 36 values held, <= 4 waves / SIMD       %-20s %8.1f us  %6.2f TB/s (writes)\n", "kernel layout", t, vbytes / t / 1e6);
  t = run4<true>(V, x, nt, C, iters);    printf("  the same + the kernel's vector work (72 SiLU, ~700 ops) %-16s %8.1f us  %6.2f TB/s (writes)\n", "kernel layout", t, vbytes / t / 1e6);
  return 0;
}
