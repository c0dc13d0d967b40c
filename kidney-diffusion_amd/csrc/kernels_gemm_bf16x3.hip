// fp32 GEMMs on the bf16 matrix pipe: every fp32 operand is carried as three bf16 pieces,
//
//   a = ah + am + al       ah = bf16(a), am = bf16(a - ah), al = bf16(a - ah - am)     (each subtraction is exact;
//                          3 x 8 significant bits hold the 24 of an fp32 mantissa),
//   a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + terms <= 2^-24 |a b|,
//
// six v_mfma_f32_32x32x16_bf16 per 16-deep k-step, every product exact, accumulated in fp32 from the smallest terms up.
// The result is as close to the exact product as the fp32 MFMA's (measured against fp64 on the shapes below: rms error
// 3.4e-7 .. 1.4e-6 of the result's rms, fp32 MFMA 4.1e-7 .. 1.6e-6; scratch/bf16x3/, profiles/README.md "bf16x3") -
// this is fp32 arithmetic issued on a faster pipe, not a reduced precision: the bf16 pipe issues 16 x the MACs per clock
// of v_mfma_f32_32x32x2_f32, so six products cost 3/8 of the fp32 instruction's matrix cycles.
//
// Used for the 36 position GEMMs of Winograd F(4x4,3x3) (kernels_wino4.hip), D_g[t][n] = sum_c V_g[t][c] U_g[n][c]:
//
//   operands   three planes in K-CHUNK-MAJOR order, [3][G][K/16][R][16] bf16: the 32 rows x 32 bytes one LDS-DMA instruction
//              moves are 1 KB of consecutive memory.  V is written in this form by the input transform (wino4_in3), U once
//              per plan (split3).
//   kernel     256 x 128 output tile, twelve waves: eight compute (64 x 64 each = 2 x 2 MFMA tiles), four only issue the
//              buffer_load ... lds of the ring of four 36 KB stages (BK = 16), three stages ahead.  (A wave whose DMA
//              waits for a slot in the memory pipeline cannot issue MFMAs meanwhile; with loaders of their own the
//              K = 512 launches took 0.41 instead of 0.51 ms.)  The fragments of stage i + 1 are read while the 24 MFMAs
//              of stage i run (two register sets); one barrier per stage.  The two 16-byte slots of a 32-byte row are
//              swapped in rows 8..15 of every 16, on the source side of the DMA and in the read: conflict-free
//              ds_read_b128 fragments.  Workgroup ids are mapped so that every XCD works through one contiguous eighth
//              of the (g, M tile, N tile) order: an operand tile crosses the fabric into one L2.
//   bound      the matrix pipe at the clock the chip sustains under bf16 MFMA load: 1.24-1.37 PFLOP/s of bf16 issued
//              (2.5 nominal), 190-230 TFLOP/s of fp32-equivalent work against 113-154 of the fp32 MFMA path.
#include "common.h"

namespace kd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = X3_BK;
constexpr int ROWB = BK * 2;                          // 32 bytes per LDS row
constexpr int NST = 4;
constexpr int STAGE_B = 3 * (BM + BN) * ROWB;         // 36 864 bytes
constexpr int B_OFF = 3 * BM * ROWB;

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

}  // namespace

// planes[p][g][k / 16][r][k % 16], p = 0 (high), 1 (middle), 2 (low), of x [G][R][K].  One thread: two consecutive k
// (v_cvt_pk_bf16_f32 rounds to nearest even)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, uint32_t* __restrict__ planes, int64_t n2,
                                                     int R, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const int K2 = K >> 1;
  const int k = (int)(i % K2) * 2;
  const int64_t gr = i / K2;
  const int r = (int)(gr % R);
  const int64_t g = gr / R;
  const int64_t o = (((g * (K / BK) + k / BK) * R + r) * BK + k % BK) >> 1;
  uint32_t h, m, l;
  x3_split(*(const f32x2*)(x + 2 * i), h, m, l);
  planes[o] = h;
  planes[n2 + o] = m;
  planes[2 * n2 + o] = l;
}

int launch_split3(const float* x, void* planes, int G, int R, int K, hipStream_t s) {
  KD_REQUIRE(K % BK == 0 && ((uintptr_t)x & 7) == 0, "bf16x3 planes need K % 16 == 0");
  const int64_t n2 = (int64_t)G * R * K / 2;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, x, (uint32_t*)planes, n2, R, K);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// A3 [3][G][K/16][M][16], B3 [3][G][K/16][N][16] bf16; C [G][M][N] fp32.  Grid: 1-D, (M / 256) (N / 128) G workgroups of 768
__global__ __launch_bounds__(768) void gemm_bf16x3_kernel(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int G, int M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtiles = M / BM, ntiles = N / BN;
  // XCD x (workgroup ids x, x + 8, ...) takes the x-th eighth of the (g, M tile, N tile) order
  unsigned n = blockIdx.x;
  if ((gridDim.x & 7) == 0) n = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nt = n % ntiles, mt = (n / ntiles) % mtiles, g = n / (ntiles * mtiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = K / BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  if (wave >= 8) {
    // loader l moves pieces 9 l .. 9 l + 8 of the 36 of every stage (0..23: A plane id / 8, rows 32 (id % 8); 24..35: B
    // plane (id - 24) / 4, rows 32 ((id - 24) % 4)).  A piece = 32 rows x 32 bytes = 1 KB of consecutive memory: lane ->
    // LDS (row = lane >> 1, slot = lane & 1), read from source slot (lane & 1) ^ ((row >> 3) & 1) of the same row
    const int l = wave - 8;
    const uint32_t planeA = (uint32_t)((int64_t)G * M * K * 2), planeB = (uint32_t)((int64_t)G * N * K * 2);
    const uint32_t gA = (uint32_t)((int64_t)g * M * K * 2), gB = (uint32_t)((int64_t)g * N * K * 2);
    const i32x4 rsA = make_rsrc(A3, 3u * planeA), rsB = make_rsrc(B3, 3u * planeB);
    const uint32_t voff = (uint32_t)((lane >> 1) * ROWB + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
    const uint32_t chunkA = (uint32_t)(M * ROWB), chunkB = (uint32_t)(N * ROWB);
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
    for (int kc = 0; kc < nk; ++kc) {   // the barriers of the computing waves' stages
      if (kc + 2 < nk) wait_vm<9>(); else wait_vm<0>();   // stage kc + 1 landed (stage kc + 2 may be in flight)
      __builtin_amdgcn_s_barrier();
      if (kc + 3 < nk) issue_stage(kc + 3);   // into the buffer of stage kc - 1, whose fragments were read before stage kc - 1 ran
    }
    return;
  }
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves of 64 x 64
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
  // one stage: barrier (stage kc + 1 landed, the loaders may refill the buffer of stage kc - 1), fragments of stage
  // kc + 1 into `nxt`, the MFMAs of stage kc on `cur`
  auto stage = [&](int kc, const Frags& cur, Frags& nxt) {
    __builtin_amdgcn_s_barrier();
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};   // am bm, al bh, ah bl, am bh, ah bm, ah bh
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[PA[t]][i], cur.b[PB[t]][j], acc[i][j], 0, 0, 0);
      if (t == 0) {
        // behind the first MFMAs, so that the wait the compiler puts before them (for `cur`, read one stage ago) does
        // not also wait for the reads of `nxt`
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nk) read_frags(nxt, kc + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
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

bool gemm_bf16x3_ok(int G, int64_t M, int N, int K) {
  return G > 0 && M > 0 && M % BM == 0 && N % BN == 0 && K % 32 == 0 && (int64_t)3 * G * M * K * 2 < ((int64_t)1 << 32) &&
         (int64_t)3 * G * N * K * 2 < ((int64_t)1 << 32) && (M / BM) * (int64_t)(N / BN) * G < 0x7fffffff;
}

int launch_gemm_bf16x3(const void* A3, const void* B3, float* C, int G, int M, int N, int K, hipStream_t s) {
  KD_REQUIRE(gemm_bf16x3_ok(G, M, N, K), "bf16x3 GEMM needs M % 256 == 0, N % 128 == 0, K % 32 == 0 and operand planes < 4 GB");
  KD_REQUIRE((((uintptr_t)A3 | (uintptr_t)B3) & 15) == 0, "bf16x3 GEMM needs 16-byte aligned operand planes");
  const dim3 grid((unsigned)((M / BM) * (N / BN) * G));
  hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(768), 0, s, (const uint16_t*)A3, (const uint16_t*)B3, C, G, M, N, K);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
