// EXPERIMENT (round 4, not part of the product): fp32 GEMM through three bf16 pieces per operand.
//
//   a = ah + am + al (each a bf16: 3 x 8 significant bits = the 24 of an fp32 mantissa), same for b;
//   a b ~ ah bh + ah bm + am bh + ah bl + al bh + am bm      (the dropped terms are <= 2^-24 |a b|)
//
// Six v_mfma_f32_32x32x16_bf16 per 16 deep k-step, products exact, accumulation in fp32: the "3xTF32" construction on
// the bf16 matrix pipe, which issues 16x the FLOPs of v_mfma_f32_32x32x2_f32 per cycle - 2.67x fewer matrix cycles
// for fp32-class results.  This file measures what an UNTUNED kernel of that kind delivers on the shapes of the
// engine's batched Winograd GEMMs (C[M][N] = A[M][K] B[N][K]^T), next to the fp32-MFMA kernel of the product.
//
//   split:  fp32 [R][K] -> three bf16 planes [3][R][K]
//   gemm:   256 x 128 tile, 8 waves (wave = 64 x 64 = 2 x 2 MFMA tiles), BK = 32, two LDS stages (144 KB), operands by
//           buffer_load ... lds with the 16-byte slots of a 64-byte row XOR-swizzled by (row >> 2) & 3 on the source
//           and on the read (conflict-free ds_read_b128 fragments)
//
// build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC gemm_bf16x3.hip -o libbf16x3.so ; driver: bf16x3_bench.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint16_t bf16_rn(float v) {   // round to nearest even (finite inputs)
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// planes[p][r][k], p = 0 (high), 1 (middle), 2 (low)
extern "C" __global__ void split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i];
  const uint16_t h = bf16_rn(a);
  const float r1 = a - bf16_f(h);
  const uint16_t m = bf16_rn(r1);
  const float r2 = r1 - bf16_f(m);
  planes[i] = h;
  planes[n + i] = m;
  planes[2 * n + i] = bf16_rn(r2);
}

constexpr int BM = 256, BN = 128, BK = 32;           // BK bf16 = 64 bytes per row
constexpr int ROWB = BK * 2;                          // bytes per LDS row
constexpr int STAGE_B = 3 * (BM + BN) * ROWB;         // 73 728 bytes

__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// A3 [3][G][M][K], B3 [3][G][N][K] bf16; C [G][M][N] fp32 (G independent products, blockIdx.z - the 36 Winograd
// positions).  M % 256 == 0, N % 128 == 0, K % 32 == 0, 3 planes < 4 GB.
// NPROD = 6 (all terms), 3 (hh + hm + mh, ~2^-16), 1 (plain bf16): the accuracy / speed ladder
template <int NPROD>
__global__ __launch_bounds__(512) void gemm_bf16x3_kernel(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves of 64 x 64
  // N tiles of one M tile back to back on one XCD
  int mt = blockIdx.x, nt = blockIdx.y;
  if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
    const unsigned lid = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned slot = lid >> 3;
    nt = slot % gridDim.y;
    mt = (slot / gridDim.y) * 8 + (lid & 7);
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int G = gridDim.z, g = blockIdx.z;
  const uint32_t planeA = (uint32_t)((int64_t)G * M * K * 2), planeB = (uint32_t)((int64_t)G * N * K * 2);
  const uint32_t gA = (uint32_t)((int64_t)g * M * K * 2), gB = (uint32_t)((int64_t)g * N * K * 2);
  const i32x4 rsA = make_rsrc(A3, 3u * planeA), rsB = make_rsrc(B3, 3u * planeB);
  // a DMA piece = 16 rows x 64 bytes: lane -> (row = lane >> 2, slot = lane & 3), source slot swizzled
  const int prow = lane >> 2, pslot = (lane & 3) ^ ((prow >> 2) & 3);
  // pieces per stage: A 3 planes x 16 pieces, B 3 planes x 8 pieces = 72; wave w carries pieces w, w + 8, ...
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  auto issue = [&](int kc, int stage) {
    const uint32_t koff = (uint32_t)(kc * ROWB);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int piece = wave + 8 * q;                  // 0..71
      const bool isA = piece < 48;
      const int pl = isA ? piece / 16 : (piece - 48) / 8;
      const int pr = isA ? piece % 16 : (piece - 48) % 8;
      const int row = pr * 16 + prow;
      const uint32_t dst = lds0 + (uint32_t)(stage * STAGE_B + (isA ? 0 : 3 * BM * ROWB) + (pl * (isA ? BM : BN) + pr * 16) * ROWB);
      const uint32_t voff = isA ? (uint32_t)pl * planeA + gA + (uint32_t)((m0 + row) * K * 2 + pslot * 16)
                                : (uint32_t)pl * planeB + gB + (uint32_t)((n0 + row) * K * 2 + pslot * 16);
      dma16(isA ? rsA : rsB, dst, voff, koff);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  auto frag = [&](const char* base, int row, int ks) -> bf16x8 {   // 8 bf16 of row `row`, k = 16 ks + 8 fh ..
    const int slot = (2 * ks + fh) ^ ((row >> 2) & 3);
    return *(const bf16x8*)(base + row * ROWB + slot * 16);
  };
  auto compute = [&](int stage) {
    const char* sA = lds + stage * STAGE_B;
    const char* sB = sA + 3 * BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = frag(sA + p * BM * ROWB, wm * 64 + i * 32 + fr, ks);
          b[p][i] = frag(sB + p * BN * ROWB, wn * 64 + i * 32 + fr, ks);
        }
      // smallest terms first
      constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t < 6 - NPROD) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
      }
    }
  };
  const int nk = K / BK;
  issue(0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    const int st = kc & 1;
    if (kc + 1 < nk) {
      issue(kc + 1, st ^ 1);
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    compute(st);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // C/D layout of 32x32 tiles: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int col = n0 + wn * 64 + j * 32 + fr;
        C[((int64_t)g * M + row) * N + col] = acc[i][j][r];
      }
}

extern "C" int bf16x3_split(const float* x, void* planes, int64_t n, void* stream) {
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)planes, n);
  return hipGetLastError() != hipSuccess;
}
extern "C" int bf16x3_gemm(const void* A3, const void* B3, float* C, int G, int M, int N, int K, int nprod, void* stream) {
  if (M % BM || N % BN || K % BK || (int64_t)3 * G * M * K * 2 >= (int64_t)1 << 32 || (int64_t)3 * G * N * K * 2 >= (int64_t)1 << 32) return 2;
  const dim3 grid(M / BM, N / BN, G), block(512);
  const uint16_t *a = (const uint16_t*)A3, *b = (const uint16_t*)B3;
  hipStream_t s = (hipStream_t)stream;
  if (nprod == 6) hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, block, 0, s, a, b, C, M, N, K);
  else if (nprod == 3) hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, grid, block, 0, s, a, b, C, M, N, K);
  else if (nprod == 1) hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, grid, block, 0, s, a, b, C, M, N, K);
  else return 2;
  return hipGetLastError() != hipSuccess;
}
