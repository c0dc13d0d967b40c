// Fused Winograd F(4x4,3x3) + GroupNorm / FiLM / SiLU: the ResnetBlock 3x3 convs with Cin 128 .. 512 on the big maps.
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T act(d) B) ] A     (interpolation points 0, +-1, +-2, inf: kernels_wino4.hip)
//
// 36 MACs per 4x4 outputs instead of 16 per 2x2 (F(2x2,3x3), kernels_wino_fused128.hip): 1.78x fewer MFMAs.  The batched
// form of kernels_wino4.hip pays for that with V and D round trips through HBM (2.25x the map each way), which is why it
// loses below Cin = 512 (profiles/README.md); here V lives in LDS and the 36 positions of a tile never leave the
// registers:
//
//   * item = 16 x 32 output pixels (4 x 8 tiles of 4 x 4) x 64 output channels, one per persistent workgroup turn;
//     K is walked in chunks of 4 input channels;
//   * 8 waves (512 threads, two per SIMD, <= 256 VGPRs): wave (mb, cb) owns tiles 16 mb .. 16 mb + 15 x output
//     channels 16 cb .. 16 cb + 15 at ALL 36 positions: 36 accumulators of v_mfma_f32_16x16x4_f32 (144 registers),
//     one MFMA per position and chunk.  A lane ends with the whole 6 x 6 transformed tile of its (tile, channel) pairs,
//     so the output transform A^T m A is register arithmetic: no exchange between waves;
//   * per chunk: the raw 18 x 34 x 4 patch arrives by buffer_load ... lds (pixel slots ordered by column residue mod 4,
//     so that the transform's reads are bank-conflict free), waves 6-7 apply GroupNorm / FiLM / SiLU in place (padding
//     pixels forced back to 0), waves 0-5 each compute ONE row i of B^T d B for all 32 tiles x 4 channels (the row of
//     B^T d is a 4-term combination of patch rows with wave-uniform coefficients) and store it as MFMA operands,
//     U (36 positions x 64 channels x 4) arrives by DMA in the same [position / 4][channel][row][position % 4] order:
//     one ds_read_b128 feeds four MFMAs per operand;
//   * stages: raw x 3, U x 2, V x 2 (145 KB of LDS); one barrier per chunk: iteration c issues raw(c+3) and U(c+1),
//     runs the MFMAs of chunk c, transforms chunk c+1 and activates chunk c+2.
#include "common.h"

#include <type_traits>

namespace kd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr uint32_t F4_OOB = 0x80000000u;   // buffer offset past num_records: the DMA writes zeros
constexpr int F4_K = 4;                    // input channels per chunk
constexpr int F4_TILES = 32;               // 4 x 8 tiles of 4 x 4 outputs
constexpr int F4_N = 64;                   // output channels per item
constexpr int F4_U = 9 * 4 * F4_N * 4;     // floats of one U chunk  [p/4][k][n][p%4]
constexpr int F4_V = 9 * 4 * F4_TILES * 4; // floats of one V chunk  [p/4][k][tile][p%4]
constexpr int F4_ROW = 34;                 // pixel slots per patch row
constexpr int F4_SLOTS = 18 * F4_ROW;      // 612
constexpr int F4_RAW = 640 * 4;            // floats of one raw stage (10 DMA pieces of 64 pixel slots)
constexpr int F4_MAXC = 512;               // channels of the affine table kept in LDS
// LDS map (floats)
constexpr int L_U0 = 0, L_U1 = F4_U, L_V0 = 2 * F4_U, L_V1 = L_V0 + F4_V, L_R0 = L_V1 + F4_V, L_R1 = L_R0 + F4_RAW,
              L_R2 = L_R1 + F4_RAW, L_AB = L_R2 + F4_RAW, L_END = L_AB + 2 * F4_MAXC;
static_assert(L_END * 4 <= 160 * 1024, "LDS");

typedef int i32x4 __attribute__((ext_vector_type(4)));
// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KB of LDS at byte address lds_addr) as inline assembly.  Through the
// builtin, hipcc orders later LDS accesses of the kernel behind the DMA with s_waitcnt vmcnt(0) wherever it cannot prove
// them apart, i.e. it exposes the whole latency right behind the issue; the kernel orders its DMAs itself (one
// s_waitcnt vmcnt(0) + barrier per chunk).  m0 is not live across this statement: nothing else in the kernel uses it.
// NOTE: any VGPR spill inside the chunk loop brings the same stall back (a scratch reload waits vmcnt(0)).
__device__ __forceinline__ void f4_dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ i32x4 f4_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// B^T applied to six values (one row of the 6 x 6 tile): the formulas of kernels_wino4.hip
__device__ __forceinline__ void f4_bt(const float (&d)[6], float (&t)[6]) {
  t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
  t[1] = (d[3] + d[4]) - 4.0f * (d[1] + d[2]);
  t[2] = (d[4] - d[3]) + 4.0f * (d[1] - d[2]);
  t[3] = (d[4] - d[2]) + 2.0f * (d[3] - d[1]);
  t[4] = (d[4] - d[2]) + 2.0f * (d[1] - d[3]);
  t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
// A^T applied to six values -> four
__device__ __forceinline__ void f4_at(const float (&m)[6], float (&y)[4]) {
  const float a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], d = m[3] - m[4];
  y[0] = m[0] + a + c;
  y[1] = b + 2.0f * d;
  y[2] = a + 4.0f * c;
  y[3] = b + 8.0f * d + m[5];
}
}  // namespace

// OIHW 3x3 weights -> U = G g G^T in the order the kernel's DMA reads: [N/64][C/4] chunks of [p/4][k][n][p%4]
__global__ __launch_bounds__(256) void wino4_fused_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int N, int C,
                                                               float scale) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * C) return;
  const int n = (int)(idx / C), c = (int)(idx % C);
  const float* g = w + idx * 9;
  float t[6][3];   // G g
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float g0 = g[k], g1 = g[3 + k], g2 = g[6 + k];
    t[0][k] = g0 * 0.25f;
    t[1][k] = -(g0 + g1 + g2) * (1.0f / 6.0f);
    t[2][k] = -(g0 - g1 + g2) * (1.0f / 6.0f);
    t[3][k] = g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
    t[4][k] = g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
    t[5][k] = g2;
  }
  const int nchunks = C / F4_K;
  float* dst = U + ((int64_t)(n / F4_N) * nchunks + c / F4_K) * F4_U;
  const int nn = n % F4_N, k = c % F4_K;
#pragma unroll
  for (int r = 0; r < 6; ++r) {   // (G g) G^T
    const float a0 = t[r][0], a1 = t[r][1], a2 = t[r][2];
    const float u[6] = {a0 * 0.25f,
                        -(a0 + a1 + a2) * (1.0f / 6.0f),
                        -(a0 - a1 + a2) * (1.0f / 6.0f),
                        a0 * (1.0f / 24.0f) + a1 * (1.0f / 12.0f) + a2 * (1.0f / 6.0f),
                        a0 * (1.0f / 24.0f) - a1 * (1.0f / 12.0f) + a2 * (1.0f / 6.0f),
                        a2};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int p = r * 6 + s;
      dst[(((p >> 2) * 4 + k) * F4_N + nn) * 4 + (p & 3)] = u[s] * scale;
    }
  }
}

// id -> (image, y0, x0, 64-channel slab); the N/64 items of one patch back to back on ONE XCD (ids go round-robin over
// the 8 XCDs)
__global__ __launch_bounds__(256) void wino4_fused_items_kernel(int4* __restrict__ out, int B, int H, int W, int N) {
  const int pw = W / 32, ph = H / 16, nh = N / F4_N;
  const int npatch = B * pw * ph;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= npatch * nh) return;
  int bpatch, slab;
  if ((npatch & 7) == 0) {
    bpatch = (id / (8 * nh)) * 8 + (id & 7);
    slab = (id >> 3) % nh;
  } else {
    bpatch = id / nh;
    slab = id % nh;
  }
  const int b = bpatch / (pw * ph), prem = bpatch - b * pw * ph;
  out[id] = make_int4(b, (prem / pw) * 16, (prem % pw) * 32, slab);
}

__global__ __launch_bounds__(512, 2) void wino4_fused_gn_kernel(const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ ab, const float* __restrict__ U,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ res, int ldres,
                                                               float* __restrict__ y, int B, int H, int W, int C, int N,
                                                               double* __restrict__ opart, const int4* __restrict__ items,
                                                               int xflags) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef KD_EXPERIMENT   // ablations (timing only, wrong results): 1 no stores, 2 no MFMAs, 4 no vector work, 8 no DMA,
  const int flags = xflags;   // 16 no operand reads, 32 no activation, 64 no transform
#else
  constexpr int flags = 0;
#endif
  // Separate LDS objects and compile-time stage indices (the chunk loop is unrolled six times): one array with run-time
  // stage offsets costs address arithmetic per access
  __shared__ __attribute__((aligned(1024))) float raw_0[F4_RAW], raw_1[F4_RAW], raw_2[F4_RAW];
  __shared__ __attribute__((aligned(1024))) float us_0[F4_U], us_1[F4_U];
  __shared__ __attribute__((aligned(1024))) float vs[2 * F4_V];   // both V stages; the epilogue's exchange tile
  __shared__ __attribute__((aligned(16))) float abl[2 * F4_MAXC];
  using LP = __attribute__((address_space(3))) float*;
  auto rawp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return raw_0; else if constexpr (decltype(S)::value == 1) return raw_1; else return raw_2; };
  auto usp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return us_0; else return us_1; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave >> 2, cb = wave & 3;   // (the four waves of a tile half sit on the four SIMDs)
  const int pw = W / 32, ph_ = H / 16;
  const int nitems = B * pw * ph_ * (N / F4_N);
  const int nchunks = C / F4_K;
  const i32x4 rsU = f4_rsrc(U, (uint32_t)((int64_t)36 * N * C * 4));

  // ---- MFMA role: A = V[p][16 mb + (lane & 15)][lane >> 4], B = U[p][16 cb + (lane & 15)][lane >> 4]
  const int aoff = ((lane >> 4) * F4_TILES + mb * 16 + (lane & 15)) * 4;
  const int boff = ((lane >> 4) * F4_N + cb * 16 + (lane & 15)) * 4;

  // ---- transform role (waves 0-2): TWO rows of B^T d B for tile (ty, tx) and channels 2 chp, 2 chp + 1.  The rows
  // of B^T pair up with shared terms - (1, 2): a = d4 - 4 d2, b = d3 - 4 d1, a + b | a - b;  (3, 4): a = d4 - d2,
  // b = d3 - d1, a + 2 b | a - 2 b;  (0, 5): 4 d0 - 5 d2 + d4 | 4 d1 - 5 d3 + d5 - so the three pair tasks do the
  // 144 operations per tile and channel of the separable transform, and read each patch value 2.3 times instead of 4
  const int ttx = lane & 7, tchp = (lane >> 3) & 1, tty = lane >> 4;
  const int ttile = tty * 8 + ttx;
  const int rbase = ((4 * tty) * F4_ROW + ttx) * 4 + tchp * 2;   // patch pixel (4 ty, 4 tx), channel pair
  const float talpha = wave == 0 ? -4.f : -1.f, tbeta = wave == 0 ? 1.f : 2.f;
  const int i_odd = wave == 0 ? 1 : wave == 1 ? 3 : 5, i_even = wave == 0 ? 2 : wave == 1 ? 4 : 0;
  // V store of row i: positions 6 i .. 6 i + 5 of channel ch at [p/4][ch][tile][p%4]
  const int vw_odd = ((((6 * i_odd) >> 2) * 4 + 2 * tchp) * F4_TILES + ttile) * 4;
  const int vw_even = ((((6 * i_even) >> 2) * 4 + 2 * tchp) * F4_TILES + ttile) * 4;

  // ---- activation role (waves 3-7): float2 elements (first + r) * 64 + lane, r < nact, of the 1224 of a raw stage
  // (612 pixel slots x 2 channel pairs): 20 rounds of 64 over five waves - 5 on waves 3 and 7 (whose SIMD carries no
  // transform wave), 4 / 3 / 3 on waves 4 / 5 / 6 (SIMD partners of the transform waves 0 / 1 / 2)
  const int nact = wave < 3 ? 0 : (wave == 3 || wave == 7) ? 5 : wave == 4 ? 4 : 3;
  const int afirst = wave == 3 ? 0 : wave == 4 ? 5 : wave == 5 ? 9 : wave == 6 ? 12 : 15;
  const int aoff0 = afirst * 64 + lane;

  // ---- per-item state
  int b, y0, x0, slab, prem;
  i32x4 rsX;
  uint32_t voffA = F4_OOB, voffB = F4_OOB;   // the thread's (up to two) raw pieces
  uint32_t amask = 0;                        // activation rounds whose pixel lies inside the image
  // raw pieces: waves 4-7 carry pieces 2 (w - 4), 2 (w - 4) + 1; waves 0-1 pieces 8, 9
  const int rawp0 = wave >= 4 ? 2 * (wave - 4) : (wave < 2 ? 8 + wave : -1);
  const int rawp1 = wave >= 4 ? rawp0 + 1 : -1;
  auto slot_pixel = [&](int slot, int& iy, int& ix) {
    const int py = slot / F4_ROW, q = slot - py * F4_ROW;
    const int px = q < 9 ? 4 * q : q < 18 ? 4 * (q - 9) + 1 : q < 26 ? 4 * (q - 18) + 2 : 4 * (q - 26) + 3;
    iy = y0 - 1 + py;
    ix = x0 - 1 + px;
    return slot < F4_SLOTS && iy >= 0 && iy < H && ix >= 0 && ix < W;
  };
  auto setup = [&](int id) {
    const int4 it = items[id];
    b = it.x;
    y0 = it.y;
    x0 = it.z;
    slab = it.w;
    prem = (y0 >> 4) * pw + (x0 >> 5);
    rsX = f4_rsrc(x + (int64_t)b * H * W * ldx, (uint32_t)((int64_t)H * W * ldx * 4));
    int iy, ix;
    voffA = voffB = F4_OOB;
    if (rawp0 >= 0 && slot_pixel(rawp0 * 64 + lane, iy, ix)) voffA = (uint32_t)(((iy * W + ix) * ldx) * 4);
    if (rawp1 >= 0 && slot_pixel(rawp1 * 64 + lane, iy, ix)) voffB = (uint32_t)(((iy * W + ix) * ldx) * 4);
    amask = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (r < nact && slot_pixel((aoff0 + 64 * r) >> 1, iy, ix)) amask |= 1u << r;
  };
  auto lds_addr = [&](const float* p) { return (uint32_t)(uintptr_t)(LP)p; };
  auto issue_raw = [&](int chunk, float* stage) {
    const uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t)(chunk * F4_K * 4));
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr(stage));
    if (rawp0 >= 0) f4_dma16(rsX, la + (uint32_t)(rawp0 * 1024), voffA, sx);
    if (rawp1 >= 0) f4_dma16(rsX, la + (uint32_t)(rawp1 * 1024), voffB, sx);
  };
  auto issue_u = [&](int chunk, float* stage) {   // 36 pieces of 1 KB: wave w carries 4 w .. 4 w + 3 and (w < 4) 32 + w
    const uint32_t su = __builtin_amdgcn_readfirstlane((uint32_t)(((slab * nchunks + chunk) * F4_U) * 4));
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr(stage));
    const uint32_t vo = (uint32_t)(lane * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) f4_dma16(rsU, la + (uint32_t)((wave * 4 + q) * 1024), vo, su + (uint32_t)((wave * 4 + q) * 1024));
    if (wave < 4) f4_dma16(rsU, la + (uint32_t)((32 + wave) * 1024), vo, su + (uint32_t)((32 + wave) * 1024));
  };
  // GroupNorm / FiLM / SiLU of one raw stage in place.  ab holds -log2(e) (A, B): u = -log2(e) v, e^-v = 2^u,
  // u / (1 + 2^u) = -log2(e) SiLU(v); -ln 2 sits in U (WF_U_SCALE).  All reads first, then the arithmetic, then the
  // writes: one LDS round trip per chunk, not one per round
  auto activate = [&](int chunk, float* stage, auto MASK) {
    const f32x4 a4 = *(const f32x4*)(abl + 2 * (chunk * F4_K + (lane & 1) * 2));
    f32x2* ap = (f32x2*)(stage + aoff0 * 2);
    f32x2 v[5];
    // (wave 7's last round is partly past the 1224 elements)
    auto live = [&](int r) { return r < nact && (r < 4 || wave != 7 || aoff0 + 64 * 4 < F4_SLOTS * 2); };
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (live(r)) v[r] = ap[64 * r];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const float u0 = fmaf(v[r][0], a4[0], a4[1]), u1 = fmaf(v[r][1], a4[2], a4[3]);
      v[r][0] = u0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u0));
      v[r][1] = u1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u1));
      if (decltype(MASK)::value && !((amask >> r) & 1)) v[r][0] = v[r][1] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (live(r)) ap[64 * r] = v[r];
  };
  // column c of the tile: pixel 4 tx + c -> slot offset {0, 9, 18, 26, 1, 10} (residue block, then index), in floats
  constexpr int CO[6] = {0, 9 * 4, 18 * 4, 26 * 4, 1 * 4, 10 * 4};
  constexpr int CH = F4_TILES * 4;       // channel stride of V
  constexpr int PG = 4 * F4_TILES * 4;   // position-group stride of V
  // one row of B^T d (both channels) -> B^T along the row -> positions 6 i .. 6 i + 5 of V
  auto store_even = [&](float* Vd, const float (&wa)[6], const float (&wb)[6]) {   // 6 i % 4 == 0: a full group, then two
    float ta[6], tb[6];
    f4_bt(wa, ta);
    f4_bt(wb, tb);
    *(f32x4*)(Vd) = f32x4{ta[0], ta[1], ta[2], ta[3]};
    *(f32x4*)(Vd + CH) = f32x4{tb[0], tb[1], tb[2], tb[3]};
    *(f32x2*)(Vd + PG) = f32x2{ta[4], ta[5]};
    *(f32x2*)(Vd + PG + CH) = f32x2{tb[4], tb[5]};
  };
  auto store_odd = [&](float* Vd, const float (&wa)[6], const float (&wb)[6]) {    // 6 i % 4 == 2: two, then a full group
    float ta[6], tb[6];
    f4_bt(wa, ta);
    f4_bt(wb, tb);
    *(f32x2*)(Vd + 2) = f32x2{ta[0], ta[1]};
    *(f32x2*)(Vd + CH + 2) = f32x2{tb[0], tb[1]};
    *(f32x4*)(Vd + PG) = f32x4{ta[2], ta[3], ta[4], ta[5]};
    *(f32x4*)(Vd + PG + CH) = f32x4{tb[2], tb[3], tb[4], tb[5]};
  };
  auto transform = [&](const float* raw_stage, int v_off) {
    const float* R = raw_stage + rbase;
    float oa[6], ob[6], ea[6], eb[6];   // the odd and the even row of the pair, channels a / b
    if (wave < 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c == 3) __builtin_amdgcn_sched_barrier(0);   // (12 reads in flight at a time: registers)
        const f32x2 d1 = *(const f32x2*)(R + 1 * F4_ROW * 4 + CO[c]), d2 = *(const f32x2*)(R + 2 * F4_ROW * 4 + CO[c]),
                    d3 = *(const f32x2*)(R + 3 * F4_ROW * 4 + CO[c]), d4 = *(const f32x2*)(R + 4 * F4_ROW * 4 + CO[c]);
        const float a0 = fmaf(talpha, d2[0], d4[0]), b0 = fmaf(talpha, d1[0], d3[0]);
        const float a1 = fmaf(talpha, d2[1], d4[1]), b1 = fmaf(talpha, d1[1], d3[1]);
        oa[c] = fmaf(tbeta, b0, a0);
        ea[c] = fmaf(-tbeta, b0, a0);
        ob[c] = fmaf(tbeta, b1, a1);
        eb[c] = fmaf(-tbeta, b1, a1);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c == 2 || c == 4) __builtin_amdgcn_sched_barrier(0);   // (12 reads in flight at a time, not 36: registers)
        const f32x2 d0 = *(const f32x2*)(R + CO[c]), d2 = *(const f32x2*)(R + 2 * F4_ROW * 4 + CO[c]),
                    d4 = *(const f32x2*)(R + 4 * F4_ROW * 4 + CO[c]);
        const f32x2 d1 = *(const f32x2*)(R + 1 * F4_ROW * 4 + CO[c]), d3 = *(const f32x2*)(R + 3 * F4_ROW * 4 + CO[c]),
                    d5 = *(const f32x2*)(R + 5 * F4_ROW * 4 + CO[c]);
        ea[c] = fmaf(4.0f, d0[0], fmaf(-5.0f, d2[0], d4[0]));
        eb[c] = fmaf(4.0f, d0[1], fmaf(-5.0f, d2[1], d4[1]));
        oa[c] = fmaf(4.0f, d1[0], fmaf(-5.0f, d3[0], d5[0]));
        ob[c] = fmaf(4.0f, d1[1], fmaf(-5.0f, d3[1], d5[1]));
      }
    }
    store_odd(vs + v_off + vw_odd, oa, ob);
    store_even(vs + v_off + vw_even, ea, eb);
  };

  f32x4 acc[36];
  auto mfmas = [&](const float* u_stage, int v_off) {
    const float* va = vs + v_off + aoff;
    const float* ub = u_stage + boff;
    // three batches of three position groups; the reads of batch k + 1 are in flight under the MFMAs of batch k
    // (batches of two: +1.5 %)
    f32x4 a0[3], b0[3], a1[3], b1[3];
    auto rd = [&](int g0, f32x4 (&a)[3], f32x4 (&bq)[3]) {
      if (flags & 16) {   // (ablation: operands without LDS reads)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const float z = (float)(g0 + g + lane);
          a[g] = f32x4{z, z + 1.f, z + 2.f, z + 3.f};
          bq[g] = f32x4{z, z - 1.f, z - 2.f, z - 3.f};
          asm volatile("" : "+v"(a[g]), "+v"(bq[g]));
        }
        return;
      }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        a[g] = *(const f32x4*)(va + (g0 + g) * (4 * F4_TILES * 4));
        bq[g] = *(const f32x4*)(ub + (g0 + g) * (4 * F4_N * 4));
      }
    };
    auto mm = [&](int g0, const f32x4 (&a)[3], const f32x4 (&bq)[3]) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int p = 4 * (g0 + g);
        acc[p + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][0], bq[g][0], acc[p + 0], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][1], bq[g][1], acc[p + 1], 0, 0, 0);
        acc[p + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][2], bq[g][2], acc[p + 2], 0, 0, 0);
        acc[p + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][3], bq[g][3], acc[p + 3], 0, 0, 0);
      }
    };
    rd(0, a0, b0);
    rd(3, a1, b1);
    mm(0, a0, b0);
    rd(6, a0, b0);
    mm(3, a1, b1);
    mm(6, a0, b0);
  };

  using MaskT = std::integral_constant<bool, true>;
  using MaskF = std::integral_constant<bool, false>;
  bool border = false;
  // the vector work of one interval: transform of chunk ct (waves 0-2) into V stage v_t, activation of chunk ca (waves 3-7)
  auto vector_work = [&](int ct, const float* raw_t, int v_t, int ca, float* raw_a) {
    if (wave < 3) {
      if (ct < nchunks && !(flags & 64)) transform(raw_t, v_t);
    } else if (ca < nchunks && !(flags & 32)) {
      if (border) activate(ca, raw_a, MaskT{});
      else activate(ca, raw_a, MaskF{});
    }
  };
  // first raw chunks and U chunk 0 of the item that setup() just described (raw and U stages are free)
  auto issue_first = [&]() {
    issue_raw(0, raw_0);
    if (1 < nchunks) issue_raw(1, raw_1);
    if (2 < nchunks) issue_raw(2, raw_2);
    issue_u(0, us_0);
  };

  int item = blockIdx.x;
  setup(item);
  issue_first();
  f32x2 abv = f32x2{0.f, 0.f};
  if (tid < C) abv = ((const f32x2*)ab)[(int64_t)b * C + tid];
  bool first = true;
  while (true) {
    // ---- top of an item: its raw chunks 0-2 and U chunk 0 are in flight (issued in front of the previous item's epilogue),
    // abv holds its affine table row.  Later items: a wave's younger operations are its 16 output stores (and the two
    // stores of the statistics): all but those 18 done means the prefetch has landed, while the stores drain under the
    // two prologue intervals and the first iteration
    if (tid < C) *(f32x2*)(abl + 2 * tid) = abv;
    border = y0 == 0 || x0 == 0 || y0 + 16 >= H || x0 + 32 >= W;
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vector_work(nchunks, raw_0, 0, 0, raw_0);       // activate chunk 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vector_work(0, raw_0, 0, 1, raw_1);             // transform chunk 0, activate chunk 1

    // iteration c = c0 + J (c0 % 6 == 0; stages of chunk j: raw j % 3, U / V j % 2): raw(c+2) and U(c) have landed,
    // barrier; DMA of raw(c+3) into the stage the transform of iteration c - 1 consumed and of U(c+1); MFMAs of chunk c,
    // transform of chunk c+1, activation of chunk c+2.  The two waves of a SIMD (w and w + 4) run the interval in
    // opposite order - vector work first on waves 0-3 (the transform waves and one activation wave), matrix work first
    // on waves 4-7.  (ONE copy of the MFMA block per body: two behind a branch make hipcc spill the accumulators.
    // Issuing the matrix-first waves' DMA pieces behind their MFMAs was measured: the second call site costs registers,
    // the loop spills, and a scratch reload is a vmcnt(0) wait behind the DMA: +12 %)
    auto body = [&](int c, auto JJ) {
      constexpr int J = decltype(JJ)::value;
      using Rc = std::integral_constant<int, J % 3>;
      using Rc1 = std::integral_constant<int, (J + 1) % 3>;
      using Rc2 = std::integral_constant<int, (J + 2) % 3>;
      using Uc = std::integral_constant<int, J % 2>;
      using Uc1 = std::integral_constant<int, (J + 1) % 2>;
      constexpr int Vc = (J % 2) * F4_V, Vc1 = ((J + 1) % 2) * F4_V;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (c + 3 < nchunks && !(flags & 8)) issue_raw(c + 3, rawp(Rc{}));
      if (c + 1 < nchunks && !(flags & 8)) issue_u(c + 1, usp(Uc1{}));
      const bool vfirst = wave < 4;
      if (vfirst && !(flags & 4)) vector_work(c + 1, rawp(Rc1{}), Vc1, c + 2, rawp(Rc2{}));
      if (!(flags & 2)) mfmas(usp(Uc{}), Vc);
      if (!vfirst && !(flags & 4)) vector_work(c + 1, rawp(Rc1{}), Vc1, c + 2, rawp(Rc2{}));
    };
    for (int c = 0; c < nchunks; c += 6) {
#define KD_G(J) \
  if (c + J < nchunks) body(c + J, std::integral_constant<int, J>{});
      KD_G(0) KD_G(1) KD_G(2) KD_G(3) KD_G(4) KD_G(5)
#undef KD_G
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every stage is free from here on

    // ---- this item's coordinates for the epilogue; the loader state moves on to the next item, whose first three raw
    // chunks and first U chunk start now (the output exchange below lives in the V stages) and whose affine row is fetched
    const int eb = b, ey0 = y0, ex0 = x0, eslab = slab, eprem = prem;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;
    if (has_next) {
      setup(next);
      issue_first();
      if (tid < C) abv = ((const f32x2*)ab)[(int64_t)b * C + tid];
    }

    // ---- output transform Y = A^T m A in registers (the lane holds positions 0..35 of tiles 16 mb + 4 (lane >> 4) + r at
    // output channel 16 cb + (lane & 15)), then a turn through LDS so that every pixel's 64 channels (256 bytes) leave
    // in 16-byte pieces: X[8 tiles][16 pixels][64 channels] = 32 KB over the V stages, four rounds: in round h every
    // wave transforms its accumulator row h (tiles 16 mb + 4 (lane >> 4) + h); all eight waves read the round back (16
    // pixels each), add bias / residual and store
    float* const X = vs;
    const int n4 = eslab * F4_N + (lane & 15) * 4;
    const f32x4 b4 = bias ? *(const f32x4*)(bias + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* const yb = y + (int64_t)eb * H * W * N;
    const float* const rb = (res && !(flags & 1)) ? res + (int64_t)eb * H * W * ldres : nullptr;
    float fs1 = 0.f, fs2 = 0.f;
    // round h: accumulator row r = h of EVERY wave: X tile (mb * 4 + (lane >> 4)) = tile 16 mb + 4 (lane >> 4) + h
    auto out_row = [&](auto RR) {
      constexpr int r = decltype(RR)::value;
      float* xw = X + ((mb * 4 + (lane >> 4)) * 16) * F4_N + cb * 16 + (lane & 15);
      float sc[4][6];   // A^T m: columns first
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float m[6] = {acc[j][r], acc[6 + j][r], acc[12 + j][r], acc[18 + j][r], acc[24 + j][r], acc[30 + j][r]};
        float o[4];
        f4_at(m, o);
#pragma unroll
        for (int a = 0; a < 4; ++a) sc[a][j] = o[a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float o[4];
        f4_at(sc[a], o);
#pragma unroll
        for (int q = 0; q < 4; ++q) xw[(a * 4 + q) * F4_N] = o[q];
      }
    };
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      // pixel of the round this lane reads back: P = wave * 16 + j * 4 + (lane >> 4): X tile xt = P >> 4 is tile
      // 16 (xt >> 2) + 4 (xt & 3) + h
      f32x4 rv[4];
      uint32_t pixv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int P = wave * 16 + j * 4 + (lane >> 4);
        const int xt = P >> 4;
        const int t = (xt >> 2) * 16 + (xt & 3) * 4 + h;
        pixv[j] = (uint32_t)((ey0 + 4 * (t >> 3) + ((P >> 2) & 3)) * W + ex0 + 4 * (t & 7) + (P & 3));
        if (rb) rv[j] = *(const f32x4*)(rb + (pixv[j] * (uint32_t)ldres + (uint32_t)n4));   // in flight under the transform
      }
      if (h == 0) out_row(std::integral_constant<int, 0>{});
      else if (h == 1) out_row(std::integral_constant<int, 1>{});
      else if (h == 2) out_row(std::integral_constant<int, 2>{});
      else out_row(std::integral_constant<int, 3>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int P = wave * 16 + j * 4 + (lane >> 4);
        f32x4 v = *(const f32x4*)(X + P * F4_N + (lane & 15) * 4);
        v += b4;
        if (rb) v += rv[j];
        if (!(flags & 1) || v[0] == 123.456f) *(f32x4*)(yb + (pixv[j] * (uint32_t)N + (uint32_t)n4)) = v;
        fs1 += (v[0] + v[1]) + (v[2] + v[3]);
        fs2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], fs2))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // the round has been read: the next round (or the next item's V) may overwrite it
    }
    if (opart) {   // the lane's 4 channels lie in 16-channel segment (lane & 15) >> 2: one entry per (patch, wave, segment)
      double gs1 = (double)fs1, gs2 = (double)fs2;
#pragma unroll
      for (int off = 1; off <= 2; off <<= 1) {
        gs1 += __shfl_xor(gs1, off, 64);
        gs2 += __shfl_xor(gs2, off, 64);
      }
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        gs1 += __shfl_xor(gs1, off, 64);
        gs2 += __shfl_xor(gs2, off, 64);
      }
      if ((lane & 0x33) == 0) {
        const int seg = eslab * (F4_N / 16) + (lane >> 2);
        const int64_t nchunk = (int64_t)pw * ph_ * 8;
        double* op = opart + ((((int64_t)eb * (N / 16) + seg) * nchunk) + (int64_t)eprem * 8 + wave) * 2;
        op[0] = gs1;
        op[1] = gs2;
      }
    }
    if (!has_next) break;
    item = next;
    first = false;
  }
#endif
}

bool wino4_fused_ok(int B, int H, int W, int C, int N) {
  return B > 0 && H >= 16 && W >= 32 && H % 16 == 0 && W % 32 == 0 && C >= 8 && C % F4_K == 0 && C <= F4_MAXC && N >= F4_N &&
         N % F4_N == 0 && (int64_t)H * W * C * 4 < 0x7fffffff && (int64_t)36 * N * C * 4 < 0x7fffffff &&
         (int64_t)H * W * N * 4 < 0x7fffffff && (int64_t)B * (H / 16) * (W / 32) * (N / F4_N) < 0x7fffffff;
}

int launch_wino4_fused_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s, float scale) {
  KD_REQUIRE(O % F4_N == 0 && I % F4_K == 0, "fused F(4x4,3x3) weights need Cout % 64 == 0 and Cin % 4 == 0");
  const int64_t total = (int64_t)O * I;
  hipLaunchKernelGGL(wino4_fused_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, U, O, I, scale);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t wino4_fused_items_count(int B, int H, int W, int N) { return (size_t)B * (H / 16) * (W / 32) * (N / F4_N); }
// chunks per image and 16-channel segment of the statistics the epilogue leaves: one per (patch, wave)
size_t wino4_fused_out_stats_chunks(int H, int W) { return (size_t)(H / 16) * (W / 32) * 8; }

int launch_wino4_fused_items(void* items, int B, int H, int W, int N, hipStream_t s) {
  const size_t n = wino4_fused_items_count(B, H, W, N);
  hipLaunchKernelGGL(wino4_fused_items_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int4*)items, B, H, W, N);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_wino4_fused_gn(const float* x, int ldx, const float* ab, const float* U, const float* bias, const float* res,
                          int ldres, float* y, int B, int H, int W, int C, int N, double* out_partial, const void* items,
                          hipStream_t s) {
  KD_REQUIRE(ldx >= C && ldx % 4 == 0 && (int64_t)H * W * ldx * 4 < 0x7fffffff && ((uintptr_t)x & 15) == 0,
             "GroupNorm-fused F(4x4,3x3) conv: bad input row stride");
  KD_REQUIRE(wino4_fused_ok(B, H, W, C, N),
             "GroupNorm-fused F(4x4,3x3) conv needs H % 16 == 0, W % 32 == 0, Cin % 4 == 0, Cin <= 512, Cout % 64 == 0");
  KD_REQUIRE(items != nullptr, "GroupNorm-fused F(4x4,3x3) conv: item table missing (launch_wino4_fused_items)");
  KD_REQUIRE(!res || ((int64_t)H * W * ldres * 4 < 0x7fffffff && ldres >= N), "GroupNorm-fused F(4x4,3x3) conv: bad residual");
  KD_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0 && ldres % 4 == 0,
             "GroupNorm-fused F(4x4,3x3) conv: output, bias and residual rows must be 16-byte aligned");
  const unsigned grid = (unsigned)wino4_fused_items_count(B, H, W, N);
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    KD_HIP_CHECK(hipGetDevice(&dev));
    KD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 8;   // a multiple of the 8 XCDs
  }
  const unsigned pgrid = grid < (unsigned)cus ? grid : (unsigned)cus;   // persistent: one workgroup per CU
  const int xflags = kd_switch("KD_W4F_FLAGS", 0);   // (experiment builds only)
  hipLaunchKernelGGL(wino4_fused_gn_kernel, dim3(pgrid), dim3(512), 0, s, x, ldx, ab, U, bias, res, ldres, y, B, H, W, C, N,
                     out_partial, (const int4*)items, xflags);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
