// EXPERIMENT (round 4, not part of the product): fp32 GEMM through three bf16 pieces per operand.
//
//   a = ah + am + al (each a bf16: 3 x 8 significant bits = the 24 of an fp32 mantissa), same for b;
//   a b ~ ah bh + ah bm + am bh + ah bl + al bh + am bm      (the dropped terms are <= 2^-24 |a b|)
//
// Six v_mfma_f32_32x32x16_bf16 per 16 deep k-step, products exact, accumulation in fp32: the "3xTF32" construction on
// the bf16 matrix pipe, which issues 16x the FLOPs of v_mfma_f32_32x32x2_f32 per cycle - 2.67x fewer matrix cycles
// for fp32-class results.  This file measures what such a kernel delivers on the shapes of the engine's batched
// Winograd GEMMs (C[g][M][N] = A[g][M][K] B[g][N][K]^T, g = the 36 positions), next to fp32 MFMA (torch.bmm).
//
//   split:  fp32 [G][R][K] -> three bf16 planes in K-CHUNK-MAJOR order [3][G][K/16][R][16]: the 32 rows x 32 bytes one
//           DMA instruction moves are 1 KB of consecutive memory
//   gemm:   256 x 128 tile, 8 waves (wave = 64 x 64 = 2 x 2 MFMA tiles), BK = 16 (one MFMA k-step), a ring of four LDS
//           stages (144 KB) filled by buffer_load ... lds three stages ahead; the fragments of stage i + 1 are read
//           while the 24 MFMAs of stage i run (two register sets), one barrier per stage.  The two 16-byte slots of a
//           32-byte row are swapped in rows 8..15 of every 16 (on the source side of the DMA and in the read): the
//           ds_read_b128 fragments are bank-conflict free.  Workgroups are numbered so that every XCD works through one
//           contiguous eighth of the (g, M tile, N tile) order: an operand tile is fetched over the fabric by one L2.
//
//           LOADERS template switch: the DMAs are issued by four extra waves (one per SIMD) that do nothing else - a
//           wave whose buffer_load waits for a slot in the memory pipeline cannot issue its MFMAs meanwhile, and with
//           every wave of the workgroup in the same phase behind the barrier nothing else fills the matrix pipe (v4
//           measured: time = memory time + MFMA time, not their maximum).
//
// History (profiles/README.md "bf16x3"): v1 [R][K] rows, BK = 32, two stages, all DMAs up front; v2 k-chunk-major
// operands + DMAs spread over the MFMA groups; v3 the same through registers (same time: the LDS-DMA path is not the
// limit); v4 ring of four stages, fragments one stage ahead, XCD-contiguous order; v5 loader waves.
//
// build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC gemm_bf16x3.hip -o libbf16x3.so ; driver: bf16x3_bench.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 16;
constexpr int ROWB = BK * 2;                          // 32 bytes per LDS row
constexpr int NST = 4;
constexpr int STAGE_B = 3 * (BM + BN) * ROWB;         // 36 864 bytes
constexpr int B_OFF = 3 * BM * ROWB;

// planes[p][g][k / 16][r][k % 16], p = 0 (high), 1 (middle), 2 (low);  x [G][R][K].  One thread: two consecutive k
// (v_cvt_pk_bf16_f32 rounds to nearest even; x - bf16(x) is exact)
extern "C" __global__ void split3_kernel(const float* __restrict__ x, uint32_t* __restrict__ planes, int64_t n2, int R, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const int K2 = K >> 1;
  const int k = (int)(i % K2) * 2;
  const int64_t gr = i / K2;
  const int r = (int)(gr % R);
  const int64_t g = gr / R;
  const int64_t o = (((g * (K / BK) + k / BK) * R + r) * BK + k % BK) >> 1;
  const f32x2 a = *(const f32x2*)(x + 2 * i);
  const bf16x2 h = __builtin_convertvector(a, bf16x2);
  const f32x2 r1 = a - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  planes[o] = __builtin_bit_cast(uint32_t, h);
  planes[n2 + o] = __builtin_bit_cast(uint32_t, m);
  planes[2 * n2 + o] = __builtin_bit_cast(uint32_t, l);
}

// m0 is not live across this statement (nothing else in the kernel uses it); the kernel orders its DMAs itself
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
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A3 [3][G][K/16][M][16], B3 [3][G][K/16][N][16] bf16; C [G][M][N] fp32.  M % 256 == 0, N % 128 == 0, K % 32 == 0,
// 3 planes < 4 GB.  NPROD = 6 (all terms), 3 (hh + hm + mh, ~2^-16), 1 (plain bf16): the accuracy / speed ladder.
// Grid: 1-D, (M / 256) (N / 128) G workgroups
template <int NPROD, bool LOADERS>
__global__ __launch_bounds__(LOADERS ? 768 : 512) void gemm_bf16x3_kernel(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int G, int M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves of 64 x 64
  const int mtiles = M / BM, ntiles = N / BN;
  // XCD x (workgroup ids x, x + 8, ...) takes the x-th eighth of the (g, M tile, N tile) order
  unsigned n = blockIdx.x;
  if ((gridDim.x & 7) == 0) n = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nt = n % ntiles, mt = (n / ntiles) % mtiles, g = n / (ntiles * mtiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const uint32_t planeA = (uint32_t)((int64_t)G * M * K * 2), planeB = (uint32_t)((int64_t)G * N * K * 2);
  const uint32_t gA = (uint32_t)((int64_t)g * M * K * 2), gB = (uint32_t)((int64_t)g * N * K * 2);
  const i32x4 rsA = make_rsrc(A3, 3u * planeA), rsB = make_rsrc(B3, 3u * planeB);
  // a DMA piece = 32 rows x 32 bytes = 1 KB of consecutive memory: lane -> LDS (row = lane >> 1, slot = lane & 1), read
  // from source slot (lane & 1) ^ ((row >> 3) & 1) of the same row
  const uint32_t voff = (uint32_t)((lane >> 1) * ROWB + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const uint32_t chunkA = (uint32_t)(M * ROWB), chunkB = (uint32_t)(N * ROWB);
  // pieces of a stage: A 3 planes x 8, B 3 planes x 4 = 36.  Wave w: q = 0..2 -> A plane q, rows 32 w; q = 3 -> B plane
  // w / 4, rows 32 (w % 4); q = 4 (waves 0..3 only) -> B plane 2, rows 32 w
  auto issue1 = [&](int kc, int q) {
    const int st = kc & (NST - 1);
    const bool isA = q < 3;
    const int pl = isA ? q : (q == 3 ? wave >> 2 : 2);
    const int pr = isA ? wave : (q == 3 ? wave & 3 : wave);
    const uint32_t dst = lds0 + (uint32_t)(st * STAGE_B + (isA ? 0 : B_OFF) + (pl * (isA ? BM : BN) + pr * 32) * ROWB);
    const uint32_t soff = isA ? (uint32_t)pl * planeA + gA + (uint32_t)kc * chunkA + (uint32_t)((m0 + pr * 32) * ROWB)
                              : (uint32_t)pl * planeB + gB + (uint32_t)kc * chunkB + (uint32_t)((n0 + pr * 32) * ROWB);
    dma16(isA ? rsA : rsB, dst, voff, soff);
  };
  auto issue_all = [&](int kc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue1(kc, q);
    if (wave < 4) issue1(kc, 4);
  };
  // wait until at most `behind` later stages of this wave's DMAs are in flight
  auto wait_dma = [&](int behind) {
    if (wave < 4) {
      if (behind >= 2) wait_vm<10>(); else if (behind == 1) wait_vm<5>(); else wait_vm<0>();
    } else {
      if (behind >= 2) wait_vm<8>(); else if (behind == 1) wait_vm<4>(); else wait_vm<0>();
    }
  };
  const int nk = K / BK;
  if (LOADERS && wave >= 8) {
    // loader l moves pieces 9 l .. 9 l + 8 of the 36 of every stage (0..23: A plane id / 8, rows 32 (id % 8); 24..35: B
    // plane (id - 24) / 4, rows 32 ((id - 24) % 4)); the same barriers as the computing waves
    const int l = wave - 8;
    auto issue_stage = [&](int kc) {
      const int st = kc & (NST - 1);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int id = l * 9 + q;
        const bool isA = id < 24;
        const int pl = isA ? id >> 3 : (id - 24) >> 2;
        const int pr = isA ? id & 7 : (id - 24) & 3;
        const uint32_t dst = lds0 + (uint32_t)(st * STAGE_B + (isA ? 0 : B_OFF) + (pl * (isA ? BM : BN) + pr * 32) * ROWB);
        const uint32_t soff = isA ? (uint32_t)pl * planeA + gA + (uint32_t)kc * chunkA + (uint32_t)((m0 + pr * 32) * ROWB)
                                  : (uint32_t)pl * planeB + gB + (uint32_t)kc * chunkB + (uint32_t)((n0 + pr * 32) * ROWB);
        dma16(isA ? rsA : rsB, dst, voff, soff);
      }
    };
    issue_stage(0);
    if (nk > 1) issue_stage(1);
    if (nk > 2) issue_stage(2);
    if (nk > 2) wait_vm<18>(); else if (nk > 1) wait_vm<9>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 2 < nk) wait_vm<9>(); else wait_vm<0>();   // stage kc + 1 landed (stage kc + 2 may be in flight)
      __builtin_amdgcn_s_barrier();
      if (kc + 3 < nk) issue_stage(kc + 3);
    }
    return;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  const int fslot = (fh ^ ((fr >> 3) & 1)) * 16;
  const char* fa = lds + (wm * 64 + fr) * ROWB + fslot;           // + stage, plane, 32-row block
  const char* fb = lds + B_OFF + (wn * 64 + fr) * ROWB + fslot;
  struct Frags {
    bf16x8 a[3][2], b[3][2];
  };
  auto read_frags = [&](Frags& f, int kc) {
    const int so = (kc & (NST - 1)) * STAGE_B;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f.a[p][i] = *(const bf16x8*)(fa + so + p * BM * ROWB + i * 32 * ROWB);
        f.b[p][i] = *(const bf16x8*)(fb + so + p * BN * ROWB + i * 32 * ROWB);
      }
  };
  // one stage: barrier (stage kc + 1 landed everywhere, buffer of stage kc - 1 free), DMAs of stage kc + 3, fragments
  // of stage kc + 1 into `nxt`, the MFMAs of stage kc on `cur`
  auto stage = [&](int kc, const Frags& cur, Frags& nxt) {
    if (!LOADERS && kc + 1 < nk) wait_dma(nk - 2 - kc < 1 ? nk - 2 - kc : 1);
    __builtin_amdgcn_s_barrier();
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};   // smallest terms first
#pragma unroll
    for (int t = 6 - NPROD; t < 6; ++t) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[PA[t]][i], cur.b[PB[t]][j], acc[i][j], 0, 0, 0);
      if (t == 6 - NPROD) {
        // behind the first MFMAs, so that the wait the compiler puts before them (for `cur`, read one stage ago) does
        // not also wait for the reads of `nxt`
        __builtin_amdgcn_sched_barrier(0);
        if (!LOADERS && kc + 3 < nk) issue_all(kc + 3);
        if (kc + 1 < nk) read_frags(nxt, kc + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (!LOADERS) {
    issue_all(0);
    if (nk > 1) issue_all(1);
    if (nk > 2) issue_all(2);
    wait_dma(nk - 1 < 2 ? nk - 1 : 2);
  }
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
  read_frags(f0, 0);
  for (int kc = 0; kc < nk; kc += 2) {   // nk is even (K % 32 == 0)
    stage(kc, f0, f1);
    stage(kc + 1, f1, f0);
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

extern "C" int bf16x3_split(const float* x, void* planes, int G, int R, int K, void* stream) {
  const int64_t n2 = (int64_t)G * R * K / 2;
  if (K % BK) return 2;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint32_t*)planes,
                     n2, R, K);
  return hipGetLastError() != hipSuccess;
}
extern "C" int bf16x3_gemm(const void* A3, const void* B3, float* C, int G, int M, int N, int K, int nprod, void* stream) {
  if (M % BM || N % BN || K % 32 || (int64_t)3 * G * M * K * 2 >= (int64_t)1 << 32 || (int64_t)3 * G * N * K * 2 >= (int64_t)1 << 32) return 2;
  const dim3 grid((unsigned)((M / BM) * (N / BN) * G));
  const uint16_t *a = (const uint16_t*)A3, *b = (const uint16_t*)B3;
  hipStream_t s = (hipStream_t)stream;
  if (nprod == 6) hipLaunchKernelGGL((gemm_bf16x3_kernel<6, false>), grid, dim3(512), 0, s, a, b, C, G, M, N, K);
  else if (nprod == 3) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, false>), grid, dim3(512), 0, s, a, b, C, G, M, N, K);
  else if (nprod == 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, false>), grid, dim3(512), 0, s, a, b, C, G, M, N, K);
  else if (nprod == 16) hipLaunchKernelGGL((gemm_bf16x3_kernel<6, true>), grid, dim3(768), 0, s, a, b, C, G, M, N, K);
  else if (nprod == 13) hipLaunchKernelGGL((gemm_bf16x3_kernel<3, true>), grid, dim3(768), 0, s, a, b, C, G, M, N, K);
  else if (nprod == 11) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, true>), grid, dim3(768), 0, s, a, b, C, G, M, N, K);
  else return 2;
  return hipGetLastError() != hipSuccess;
}
