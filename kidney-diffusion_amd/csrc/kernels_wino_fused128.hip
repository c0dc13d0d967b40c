// Fused Winograd F(2x2,3x3) + GroupNorm/FiLM/SiLU for the ResnetBlock 3x3 convs with Cout % 128 == 0 (every reference
// config): a sixteen-wave persistent kernel with items of 16 x 8 pixels (8 x 4 output tiles) x 128 output channels.
// (Its predecessor with items of 16 x 16 pixels x 64 channels - the same 64 accumulator registers per wave, the same
// MFMAs per chunk - served only reduced-width test models and lives in scratch/wino_fused_64ch/.)
//
// Why: on gfx950 every VALU instruction is taken from the fp32 MFMA pipe's time (profiles/README.md), and in the
// 64-channel kernel the activation (5 VALU + LDS round trip per patch value) and the input transform (8 VALU + 8 LDS
// reads + 4 stores per tile, channel and row) - both per (tile, input channel), i.e. redone for every 64-channel slab
// of the output - cost 15 % of the kernel (ablations: 19.31 ms over the 37 launches of the 64->256 UNet, 17.92
// without the activation, 17.74 without the transform, 16.36 without both).  With 128 output channels per item that
// work is done once per 128 channels: half the VALU and half the patch traffic per MFMA, for twice the U traffic
// (U is an L2 / MALL resident; the loop's DMA was 2.4 % of the kernel).
//
//   * wave (pr, wq): transformed row pr (positions 4 pr .. 4 pr + 3) x 32 tiles x output channels 32 wq .. 32 wq + 31;
//     all four waves of a row read the same V operands;
//   * transform thread = (tile, channel, row pr, half hf): columns 2 hf, 2 hf + 1 of the row: 6 patch reads, 5 VALU
//     (the row / column signs are wave-uniform multipliers of an fma, not code variants), one ds_write2;
//   * activation: 180 x 4 patch values per chunk, at most one per thread;
//   * U in 3 stages of 32 KB (chunk c + 2 is issued when chunk c - 1 has been consumed), raw patches in 4 of 4 KB, V in 2
//     of 8 KB: the loop is unrolled 12 times (static stage indices, see scratch/wino_fused_64ch/kernels_wino_fused.hip);
//   * the output transform's exchange (96 KB) lives in the three U stages, the waves' turn-around tiles (2 KB each,
//     two rounds) in U stage 2 while U chunks 0 / 1 of the next item land in stages 0 / 1; the next item's first
//     four raw chunks are fetched under the whole epilogue.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace kd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr uint32_t OOB_OFF = 0x80000000u;   // buffer offset past num_records: the DMA writes zeros
constexpr int WK = 4;                       // input channels per chunk
constexpr int W_RAW = 1024;                 // 256 pixel slots x 4 floats (18 x 10 = 180 used)
constexpr int W_U = 16 * 128 * WK;          // one chunk of U: 16 positions x 128 channels x 4
constexpr int W_V = 16 * 32 * WK;           // one chunk of V: 16 positions x 32 tiles x 4
constexpr int W_MAXC = 2048;                // channels of the affine table kept in LDS

// element (position p, row, k) of a U (ROWS = 128) or V (ROWS = 32) chunk: [p/2][k/2][row][p%2][k%2]
template <int ROWS>
__host__ __device__ constexpr int wfx_index(int p, int row, int k) {
  return ((((p >> 1) * 2 + (k >> 1)) * ROWS + row) * 4 + (p & 1) * 2 + (k & 1));
}
}  // namespace

// OIHW 3x3 weights -> U = G g G^T in the order the kernel's DMA reads: [N/128][C/4] chunks of wfx_index<128>
__global__ __launch_bounds__(256) void wino_fused128_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int N,
                                                                 int C, float scale) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * C) return;
  const int n = (int)(idx / C), c = (int)(idx % C);
  const float* g = w + idx * 9;
  float t[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float g0 = g[k], g1 = g[3 + k], g2 = g[6 + k];
    t[0][k] = g0;
    t[1][k] = 0.5f * (g0 + g1 + g2);
    t[2][k] = 0.5f * (g0 - g1 + g2);
    t[3][k] = g2;
  }
  const int nchunks = C / WK;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
    for (int s = 0; s < 4; ++s)
      U[((int64_t)(n / 128) * nchunks + c / WK) * W_U + wfx_index<128>(r * 4 + s, n % 128, c % WK)] = u[s] * scale;
  }
}

// id -> (image, y0, x0, 128-channel slab); the N/128 items of one patch back to back on ONE XCD (ids go round-robin
// over the 8 XCDs)
__global__ __launch_bounds__(256) void wino_fused128_items_kernel(int4* __restrict__ out, int B, int H, int W, int N) {
  const int pw = W / 16, ph = H / 8, nh = N / 128;
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
  out[id] = make_int4(b, (prem / pw) * 8, (prem % pw) * 16, slab);
}

__global__ __launch_bounds__(1024) void wino_fused_gn128_kernel(const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ ab,
                                                                const float* __restrict__ U,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ res, int ldres,
                                                                float* __restrict__ y, int B, int H, int W, int C, int N,
                                                                double* __restrict__ opart, int oG,
                                                                const int4* __restrict__ items, int prio) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) float raw_0[W_RAW], raw_1[W_RAW], raw_2[W_RAW], raw_3[W_RAW];
  __shared__ __attribute__((aligned(1024))) float us_0[W_U], us_1[W_U], us_2[W_U];
  __shared__ __attribute__((aligned(1024))) float vs_0[W_V], vs_1[W_V];
  __shared__ __attribute__((aligned(16))) float abl[2 * W_MAXC];
  auto rawp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return raw_0; else if constexpr (decltype(S)::value == 1) return raw_1; else if constexpr (decltype(S)::value == 2) return raw_2; else return raw_3; };
  auto usp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return us_0; else if constexpr (decltype(S)::value == 1) return us_1; else return us_2; };
  auto vsp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return vs_0; else return vs_1; };

  // tq / lq: thread and lane index as the loader, the activation and the transform see them; re-made opaque at the top
  // of every item so that what hipcc derives from them is re-derived per item, not carried through the epilogue
  int tq = threadIdx.x, lq = tq & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tq >> 6);
  const int pr = wave >> 2, wq = wave & 3;
  const int pw = W / 16, ph_ = H / 8;
  const int nitems = B * pw * ph_ * (N / 128);
  const int nchunks = C / WK;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)((int64_t)16 * N * C * 4), 0x00020000);

  // ---- per-item state (the item being LOADED: during an item's epilogue this is already the next item)
  int b, prem, y0, x0, slab;
  uint32_t voffX;
  bool aok;
  __amdgpu_buffer_rsrc_t rsX;
  // patch slot -> pixel: slot = row * 18 + (even columns 0, 2, .. 16 first, then the odd ones), rows y0 - 1 .. y0 + 8
  auto slot_pixel = [&](int slot, int& iy, int& ix) {
    int py = slot / 18, pq = slot - py * 18;
    int px = pq < 9 ? 2 * pq : 2 * (pq - 9) + 1;
    iy = y0 - 1 + py;
    ix = x0 - 1 + px;
    return slot < 180 && iy >= 0 && iy < H && ix >= 0 && ix < W;
  };
  auto setup = [&](int id) {
    const int4 it = items[id];
    b = it.x;
    y0 = it.y;
    x0 = it.z;
    slab = it.w;
    prem = (y0 >> 3) * pw + (x0 >> 4);
    rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * H * W * ldx), 0, (int)((int64_t)H * W * ldx * 4),
                                            0x00020000);
    int t_ = tq;
    asm volatile("" : "+v"(t_));
    int iy, ix;
    const bool ok = slot_pixel(t_, iy, ix);
    voffX = ok ? (uint32_t)(((iy * W + ix) * ldx) * 4) : OOB_OFF;
    aok = slot_pixel(t_ >> 2, iy, ix);   // activation: value tq of the 180 x 4 patch floats: channel tq & 3, slot tq >> 2
  };
  auto issue_raw = [&](int chunk, auto S, auto LIVE) {   // waves 0-3 only: 256 pixel slots
    __attribute__((address_space(3))) float* rb = (__attribute__((address_space(3))) float*)(rawp(S) + wave * 256);
    const uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t)(chunk * WK * 4));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, rb, 16, (decltype(LIVE)::value || chunk < nchunks) ? voffX : OOB_OFF, sx,
                                             0, 0);
  };
  auto issue_u_at = [&](int chunk, auto S, auto LIVE, uint32_t toff) {   // every wave: 2 KB of the 32 KB chunk
    __attribute__((address_space(3))) float* ub = (__attribute__((address_space(3))) float*)(usp(S) + wave * 256);
    const uint32_t su = __builtin_amdgcn_readfirstlane((uint32_t)(((slab * nchunks + chunk) * W_U) * 4));
    const bool live = decltype(LIVE)::value || chunk < nchunks;
#pragma unroll
    for (int q = 0; q < 2; ++q)   // (the second half through the SCALAR offset: one offset register for both pieces)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, ub + q * 4096, 16, live ? toff : OOB_OFF, su + (uint32_t)(q * 16384), 0, 0);
  };
  auto issue_u = [&](int chunk, auto S, auto LIVE) { issue_u_at(chunk, S, LIVE, (uint32_t)(tq * 16)); };
  auto activate = [&](int chunk, auto S, auto LIVE, auto MASK) {
    if (tq < 180 * 4) {
      const int cc = (decltype(LIVE)::value ? chunk : min(chunk, nchunks - 1)) * WK + (tq & 3);
      const float2 a2 = *(const float2*)(abl + 2 * cc);
      float* ap = rawp(S) + tq;
      // ab holds -log2(e) (A, B): u = -log2(e) v, e^-v = 2^u, u / (1 + 2^u) = -log2(e) SiLU(v); -ln 2 sits in U
      const float u = ap[0] * a2.x + a2.y;
      const float v = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
      ap[0] = (!decltype(MASK)::value || aok) ? v : 0.f;
    }
  };
  // transform thread -> offsets (set per item by derive): patch reads of rows IA / IB at the three columns, V store
  int rA, rB, voff;
  // row pr of B^T d: d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3: rows IA, IB and a sign, wave-uniform at run time
  const float sgnR = pr == 1 ? 1.0f : -1.0f;

  f32x16 acc[4];
  int aoff, boff;
  auto derive = [&]() {
    tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    lq = tq & 63;
    const int t8 = tq & 255;
    const int tc = t8 & 3, ttx = (t8 >> 2) & 7, tty = (t8 >> 5) & 3, hf = t8 >> 7;
    const int tt = tty * 8 + ttx;
    const int IA = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 2 : 1;
    const int IB = pr == 0 ? 2 : pr == 1 ? 2 : pr == 2 ? 1 : 3;
    const int roff = (2 * tty * 18 + ttx) * 4 + tc;
    rA = roff + IA * 18 * 4;
    rB = roff + IB * 18 * 4;
    voff = wfx_index<32>(pr * 4 + 2 * hf, tt, tc);
    const int frow = lq & 31, khalf = lq >> 5;
    aoff = wfx_index<32>(pr * 4, frow, khalf * 2);
    boff = wfx_index<128>(pr * 4, wq * 32 + frow, khalf * 2);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  using LiveT = std::integral_constant<bool, true>;
  using LiveF = std::integral_constant<bool, false>;
  auto run = [&](auto LD, auto HFF, auto MASK) {
    constexpr bool LOADER = decltype(LD)::value;   // waves 0-3 (row 0) carry the raw-patch DMA
    // the thread's two columns of the row (compile time: the patch reads keep immediate offsets).  Slot of column
    // 2 ttx + s relative to ttx: s = 0, 1, 2, 3 -> 0, 9, 1, 10
    //   HF = 0: v0 = u0 - u2, v1 = u1 + u2: x = u0, a = u2, w = u1, second = a + w
    //   HF = 1: v2 = u2 - u1, v3 = u1 - u3: x = u2, a = u1, w = u3, second = a - w
    constexpr int HF = decltype(HFF)::value;
    constexpr int CX = (HF ? 1 : 0) * 4, CA = (HF ? 9 : 1) * 4, CW = (HF ? 10 : 9) * 4;
    // the row of B^T d at the thread's three columns (x, a, w): held across the chunk's MFMAs as 3 values, not 6
    auto load_raw = [&](auto S, float (&u)[3]) {
      const float* ra = rawp(S) + rA;
      const float* rb = rawp(S) + rB;
      u[0] = fmaf(rb[CX], sgnR, ra[CX]);
      u[1] = fmaf(rb[CA], sgnR, ra[CA]);
      u[2] = fmaf(rb[CW], sgnR, ra[CW]);
    };
    auto write_v = [&](auto S, const float (&u)[3]) {
      float* v = vsp(S) + voff;
      v[0] = u[0] - u[1];
      v[2] = HF ? u[1] - u[2] : u[1] + u[2];
    };
    auto mfmas = [&](auto SU, auto SV) {
      const float* va = vsp(SV) + aoff;
      const float* ub = usp(SU) + boff;
      float4 a4[2], b4[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a4[i] = *(const float4*)(va + i * 2 * 32 * 4);
        b4[i] = *(const float4*)(ub + i * 2 * 128 * 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[i].x, acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[i].z, acc[2 * i + 1], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[i].y, acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[i].w, acc[2 * i + 1], 0, 0, 0);
      }
    };
    // iteration c = chunk c0 + J (c0 % 12 == 0; stages of chunk j: raw j % 4, U j % 3, V j % 2): raw(c+2) and U(c) have
    // landed (issued two iterations ago: a loader wave issues 3 pieces per iteration, the others 2), barrier, issue
    // raw(c+4) and U(c+2), MFMAs of chunk c, activate raw(c+2), transform raw(c+1) (activated one iteration ago) into V(c+1)
    auto body = [&](int c, auto JJ, auto LIVE) {
      constexpr int J = decltype(JJ)::value;
      using Rc = std::integral_constant<int, J % 4>;
      using Rc1 = std::integral_constant<int, (J + 1) % 4>;
      using Rc2 = std::integral_constant<int, (J + 2) % 4>;
      using Uc = std::integral_constant<int, J % 3>;
      using Uc2 = std::integral_constant<int, (J + 2) % 3>;
      using Vc = std::integral_constant<int, J % 2>;
      using Vc1 = std::integral_constant<int, (J + 1) % 2>;
      if constexpr (LOADER) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (LOADER) issue_raw(c + 4, Rc{}, LIVE);
      issue_u(c + 2, Uc2{}, LIVE);
      float u[3];
      load_raw(Rc1{}, u);
      mfmas(Uc{}, Vc{});
      activate(c + 2, Rc2{}, LIVE, MASK);
      write_v(Vc1{}, u);
    };
    {
      float u[3];
      load_raw(S0{}, u);
      write_v(S0{}, u);
    }
    int c = 0;
#define KD_B12(LIVE_T)                                                                                    \
  body(c, std::integral_constant<int, 0>{}, LIVE_T{});                                                    \
  body(c + 1, std::integral_constant<int, 1>{}, LIVE_T{});                                                \
  body(c + 2, std::integral_constant<int, 2>{}, LIVE_T{});                                                \
  body(c + 3, std::integral_constant<int, 3>{}, LIVE_T{});                                                \
  body(c + 4, std::integral_constant<int, 4>{}, LIVE_T{});                                                \
  body(c + 5, std::integral_constant<int, 5>{}, LIVE_T{});                                                \
  body(c + 6, std::integral_constant<int, 6>{}, LIVE_T{});                                                \
  body(c + 7, std::integral_constant<int, 7>{}, LIVE_T{});                                                \
  body(c + 8, std::integral_constant<int, 8>{}, LIVE_T{});                                                \
  body(c + 9, std::integral_constant<int, 9>{}, LIVE_T{});                                                \
  body(c + 10, std::integral_constant<int, 10>{}, LIVE_T{});                                              \
  body(c + 11, std::integral_constant<int, 11>{}, LIVE_T{});
    for (; c + 16 <= nchunks; c += 12) {   // steady state: every chunk these twelve bodies prefetch exists
      KD_B12(LiveT)
    }
#undef KD_B12
    // the last 1 .. 15 chunks: one or two trips, every body guarded (a structured `if`, not a loop exit: hipcc keeps
    // ONE copy of the accumulators)
    for (; c < nchunks; c += 12) {
#define KD_G(J) \
  if (c + J < nchunks) body(c + J, std::integral_constant<int, J>{}, LiveF{});
      KD_G(0) KD_G(1) KD_G(2) KD_G(3) KD_G(4) KD_G(5) KD_G(6) KD_G(7) KD_G(8) KD_G(9) KD_G(10) KD_G(11)
#undef KD_G
    }
  };

  // exchange messages of the output transform: (group wq, source row sr, destination index dd among the other three
  // rows) -> 2 KB (two float4 per lane); the 48 messages fill the three U stages (16 each)
  auto exmsg = [&](int sr, int dd) -> float4* {
    const int m = (wq * 4 + sr) * 3 + dd;
    float* base = m < 16 ? us_0 + m * 512 : m < 32 ? us_1 + (m - 16) * 512 : us_2 + (m - 32) * 512;
    return (float4*)base;
  };

  int item = blockIdx.x;
  derive();
  setup(item);
  // first item: the classic prologue
  if (wave < 4) {
    issue_raw(0, S0{}, LiveF{});
    issue_u(0, S0{}, LiveF{});
    issue_raw(1, S1{}, LiveF{});
    issue_raw(2, S2{}, LiveF{});
    issue_raw(3, S3{}, LiveF{});
    issue_u(1, S1{}, LiveF{});
  } else {
    issue_u(0, S0{}, LiveF{});
    issue_u(1, S1{}, LiveF{});
  }
  float2 abv[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (tq + 1024 * h < C) abv[h] = ((const float2*)ab)[(int64_t)b * C + tq + 1024 * h];
  bool first = true;
  while (true) {
    // ---- top of an item: its first raw chunks and U chunks 0 / 1 are in flight or have landed
    if (!first) derive();
    if (first) {   // (later items: written behind the previous item's exchange)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (tq + 1024 * h < C) *(float2*)(abl + 2 * (tq + 1024 * h)) = abv[h];
    }
    // first item: everything of the prologue.  Later items: the four 16-byte output stores of the previous item are the
    // youngest operations of every wave - all but 4 done means its prefetched raw 0-3, U 0 and U 1 have landed
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;
    activate(0, S0{}, LiveF{}, LiveT{});
    activate(1, S1{}, LiveF{}, LiveT{});
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool border = y0 == 0 || x0 == 0 || y0 + 8 >= H || x0 + 16 >= W;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const int hfw = (wave >> 1) & 1;   // bit 7 of the thread index
    // static issue priority for the odd rows: the four waves of a SIMD are the four rows of one wq and run the same
    // stream; a fixed priority split decides their arbitration once instead of by age every chunk (MI355X guide, two
    // waves per SIMD, item 4).  Measured over the 56 launches: -0.5 % / -1.5 % on two boxes; rows 2-3, the loader row,
    // graded priorities: no better.  KD_FWINO_PRIO=0 switches it off (A/B)
    if (prio && (pr & 1)) __builtin_amdgcn_s_setprio(1);
    if (pr == 0) {
      if (hfw == 0) {
        if (border) run(LiveT{}, H0{}, LiveT{}); else run(LiveT{}, H0{}, LiveF{});
      } else {
        if (border) run(LiveT{}, H1{}, LiveT{}); else run(LiveT{}, H1{}, LiveF{});
      }
    } else {
      if (hfw == 0) {
        if (border) run(LiveF{}, H0{}, LiveT{}); else run(LiveF{}, H0{}, LiveF{});
      } else {
        if (border) run(LiveF{}, H1{}, LiveT{}); else run(LiveF{}, H1{}, LiveF{});
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (out-of-range, zero) DMAs still write LDS
    __builtin_amdgcn_s_barrier();

    // ---- this item's coordinates for the epilogue; the loader state moves on to the next item, whose first four raw
    // chunks start now (the raw stages take no part in the exchange)
    const int eb = b, eprem = prem, ey0 = y0, ex0 = x0, en0 = slab * 128;
    int l_ = threadIdx.x & 63;   // opaque per item: the epilogue's addresses must not be hoisted over the main loop
    asm volatile("" : "+v"(l_));
    const int ti = wave * 64 + l_;
    if (has_next) {
      const int nb_img = items[next].x;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (ti + 1024 * h < C) abv[h] = ((const float2*)ab)[(int64_t)nb_img * C + ti + 1024 * h];
      setup(next);
      if (wave < 4) {
        issue_raw(0, S0{}, LiveF{});
        issue_raw(1, S1{}, LiveF{});
        issue_raw(2, S2{}, LiveF{});
        issue_raw(3, S3{}, LiveF{});
      }
    }

    // output transform Y = A^T m A.  Columns first, inside the wave: (q0, q1) = (a0 + a1 + a2, a1 - a2 - a3) of its
    // row.  Rows across the four waves of a wq group: Yrow0 = Q0 + Q1 + Q2, Yrow1 = Q1 - Q2 - Q3.  Wave pr finishes
    // the accumulator elements 4 pr .. 4 pr + 3 and hands the (q0, q1) of the other twelve to their owners
    float2 q[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      q[r] = make_float2(acc[0][r] + acc[1][r] + acc[2][r], acc[1][r] - acc[2][r] - acc[3][r]);
    double gs1 = 0.0, gs2 = 0.0;
    auto finish = [&](auto PRR) {   // the wave's row as a compile-time constant: q[] stays in registers
      constexpr int PR = decltype(PRR)::value;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (d == PR) continue;
        float4* dst = exmsg(PR, PR < d ? d - 1 : d);   // index of d among the rows other than PR
#pragma unroll
        for (int h = 0; h < 2; ++h)
          dst[h * 64 + l_] = make_float4(q[4 * d + 2 * h].x, q[4 * d + 2 * h].y, q[4 * d + 2 * h + 1].x,
                                         q[4 * d + 2 * h + 1].y);
      }
      __syncthreads();
      float2 Q[4][4];   // [source row][element i]
#pragma unroll
      for (int sr = 0; sr < 4; ++sr) {
        if (sr == PR) {
#pragma unroll
          for (int i = 0; i < 4; ++i) Q[sr][i] = q[4 * PR + i];
        } else {
          const float4* src = exmsg(sr, sr < PR ? PR - 1 : PR);   // index of PR among the rows other than sr
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 v = src[h * 64 + l_];
            Q[sr][2 * h] = make_float2(v.x, v.y);
            Q[sr][2 * h + 1] = make_float2(v.z, v.w);
          }
        }
      }
      // the exchange has been read by everybody: U chunks 0 / 1 of the next item may land on U stages 0 / 1; U stage 2
      // becomes the waves' private turn-around tiles below (2 KB each)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (has_next) {   // nobody reads the affine table before the next main loop
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (ti + 1024 * h < C) *(float2*)(abl + 2 * (ti + 1024 * h)) = abv[h];
        issue_u_at(0, S0{}, LiveF{}, (uint32_t)(ti * 16));
        issue_u_at(1, S1{}, LiveF{}, (uint32_t)(ti * 16));
      }
      // The wave's 16 outputs per lane are tile row PR of the item: pixel rows 2 PR, 2 PR + 1 x 16 pixels x 32 channels,
      // one channel per lane.  Two rounds (tile columns i = 2 h, 2 h + 1 of each lane half) through a private 2 KB of
      // LDS - [dy][xl = (2 i' + dx) + 4 half][channel] - turn them into 4 consecutive channels of one pixel per lane:
      // 16-byte stores (+ residual loads), 4 per wave instead of 16 (as its 64-channel predecessor did)
      float* sc = us_2 + wave * 512;
      const int rq = l_ >> 3, c4 = (l_ & 7) * 4;
      const int nb = en0 + wq * 32 + c4;
      const float4 b4v = bias ? *(const float4*)(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
      float* const yb = y + (int64_t)eb * H * W * N;
      const float* const rb = res + (int64_t)eb * H * W * ldres;
      float fs1 = 0.f, fs2 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float* sw = sc + (4 * (l_ >> 5)) * 32 + (l_ & 31);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * h + ii;
          sw[(2 * ii) * 32] = Q[0][i].x + Q[1][i].x + Q[2][i].x;
          sw[(2 * ii + 1) * 32] = Q[0][i].y + Q[1][i].y + Q[2][i].y;
          sw[(8 + 2 * ii) * 32] = Q[1][i].x - Q[2][i].x - Q[3][i].x;
          sw[(9 + 2 * ii) * 32] = Q[1][i].y - Q[2][i].y - Q[3][i].y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes have landed (nobody else reads them)
        // pixel column of local slot xl = rq: x = 4 h + (rq & 3) + 8 (rq >> 2)
        const uint32_t pixr = (uint32_t)((ey0 + 2 * PR) * W + ex0 + 4 * h + (rq & 3) + 8 * (rq >> 2));
#pragma unroll
        for (int k = 0; k < 2; ++k) {   // dy = k
          const uint32_t pix = pixr + (uint32_t)(k * W);
          float4 v = *(const float4*)(sc + (rq + 8 * k) * 32 + c4);
          v.x += b4v.x; v.y += b4v.y; v.z += b4v.z; v.w += b4v.w;
          if (res) {
            const float4 r4 = *(const float4*)(rb + (pix * (uint32_t)ldres + (uint32_t)nb));
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          *(float4*)(yb + (pix * (uint32_t)N + (uint32_t)nb)) = v;
          if (opart) {   // the lane's own 16 values in fp32, fp64 from there on
            fs1 += (v.x + v.y) + (v.z + v.w);
            fs2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fs2))));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next round's writes
      }
      gs1 = (double)fs1;
      gs2 = (double)fs2;
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using R3 = std::integral_constant<int, 3>;
    if (pr == 0) finish(R0{});
    else if (pr == 1) finish(R1{});
    else if (pr == 2) finish(R2{});
    else finish(R3{});
    if (opart) {   // 4 entries per item patch and 16-channel segment: row pr
      // a lane holds 4 channels 4 (l & 7) ..: lanes with (l & 7) < 4 make up segment 0 of the wave, the others segment 1
#pragma unroll
      for (int off = 8; off <= 32; off <<= 1) {
        gs1 += __shfl_xor(gs1, off, 64);
        gs2 += __shfl_xor(gs2, off, 64);
      }
#pragma unroll
      for (int off = 1; off <= 2; off <<= 1) {
        gs1 += __shfl_xor(gs1, off, 64);
        gs2 += __shfl_xor(gs2, off, 64);
      }
      if ((l_ & ~4) == 0) {   // lanes 0 and 4
        int Cg = N / oG;
        asm volatile("" : "+s"(Cg));   // (opaque: its reciprocal is not to be kept in a register across the items)
        const int cabs = en0 + wq * 32 + 4 * (l_ & 4);
        const int gg = cabs / Cg, cseg = (cabs - gg * Cg) >> 4;
        const int npi = pw * ph_;
        const int64_t chunks = (int64_t)(Cg >> 4) * npi * 4;
        const int64_t entry = ((int64_t)cseg * npi + eprem) * 4 + pr;
        double* op = opart + (((int64_t)eb * oG + gg) * chunks + entry) * 2;
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

bool wino_fused128_ok(int B, int H, int W, int C, int N) {
  return B > 0 && H >= 8 && W >= 16 && H % 8 == 0 && W % 16 == 0 && C >= WK && C % WK == 0 && C <= W_MAXC && N >= 128 &&
         N % 128 == 0 && (int64_t)H * W * C * 4 < 0x7fffffff && (int64_t)16 * N * C * 4 < 0x7fffffff &&
         (int64_t)H * W * N * 4 < 0x7fffffff && (int64_t)B * (H / 8) * (W / 16) * (N / 128) < 0x7fffffff;
}

int wino_fused_gn_max_cin() { return W_MAXC; }

// chunks per image and group of the output statistics: four (tile rows) per 16 x 8 patch and 16-channel segment
size_t wino_fused_out_stats_chunks(int H, int W, int N, int G) { return (size_t)(N / G / 16) * (H / 8) * (W / 16) * 4; }

int launch_wino_fused128_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s, float scale) {
  KD_REQUIRE(O % 128 == 0 && I % WK == 0, "fused Winograd (128-channel items) weights need Cout % 128 == 0 and Cin % 4 == 0");
  const int64_t total = (int64_t)O * I;
  hipLaunchKernelGGL(wino_fused128_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, U, O, I,
                     scale);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t wino_fused128_items_count(int B, int H, int W, int N) { return (size_t)B * (H / 8) * (W / 16) * (N / 128); }

int launch_wino_fused128_items(void* items, int B, int H, int W, int N, hipStream_t s) {
  const size_t n = wino_fused128_items_count(B, H, W, N);
  hipLaunchKernelGGL(wino_fused128_items_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int4*)items, B, H, W, N);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_wino_fused_gn128(const float* x, int ldx, const float* ab, const float* U, const float* bias, const float* res,
                            int ldres, float* y, int B, int H, int W, int C, int N, double* out_partial, int out_groups,
                            const void* items, hipStream_t s) {
  KD_REQUIRE(ldx >= C && ldx % 4 == 0 && (int64_t)H * W * ldx * 4 < 0x7fffffff && ((uintptr_t)x & 15) == 0,
             "GroupNorm-fused Winograd conv: bad input row stride");
  KD_REQUIRE(wino_fused128_ok(B, H, W, C, N),
             "GroupNorm-fused Winograd conv (128-channel items) needs H % 8 == 0, W % 16 == 0, Cin % 4 == 0, Cin <= 2048, "
             "Cout % 128 == 0");
  KD_REQUIRE(!out_partial || (out_groups > 0 && N % out_groups == 0 && (N / out_groups) % 16 == 0),
             "output statistics need groups of a multiple of 16 channels");
  KD_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0 && ldres % 4 == 0,
             "GroupNorm-fused Winograd conv: output, bias and residual rows must be 16-byte aligned");
  KD_REQUIRE(items != nullptr, "GroupNorm-fused Winograd conv: item table missing (launch_wino_fused128_items)");
  KD_REQUIRE(!res || (int64_t)H * W * ldres * 4 < 0x7fffffff, "GroupNorm-fused Winograd conv: residual images above 2 GB");
  const unsigned grid = (unsigned)wino_fused128_items_count(B, H, W, N);
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    KD_HIP_CHECK(hipGetDevice(&dev));
    KD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 8;   // a multiple of the 8 XCDs
  }
  const unsigned pgrid = grid < (unsigned)cus ? grid : (unsigned)cus;   // persistent: one workgroup per CU
  static const int prio = kd_switch("KD_FWINO_PRIO", 1);
  hipLaunchKernelGGL(wino_fused_gn128_kernel, dim3(pgrid), dim3(1024), 0, s, x, ldx, ab, U, bias, res, ldres, y, B, H, W, C, N,
                     out_partial, out_groups, (const int4*)items, prio);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
