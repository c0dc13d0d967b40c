// Fused Winograd F(2x2,3x3) convolution + GroupNorm / FiLM / SiLU for the ResnetBlock 3x3 convs whose Cout is a
// multiple of 64 but not of 128 (the layers with Cout % 128 == 0 - every layer of the reference's UNets - run the
// 128-channel items of kernels_wino_fused128.hip; this file serves reduced-width models):
//
//   y = conv3x3(SiLU(A x + B)) + bias (+ res),  x NHWC (row stride ldx), ab = per-(image, channel) affine of
//   GroupNorm (+ FiLM) (launch_gn_fold / launch_gn_fold_seg), 3x3 / stride 1 / pad 1, H % 16 == 0, W % 16 == 0,
//   Cin % 4 == 0, Cin <= 2048, Cout % 64 == 0.
//
// Nothing but x, U and y touches HBM: one work item is a 16x16-pixel patch (8x8 output tiles) x 64 output channels
// with ALL 16 Winograd positions' accumulators in registers, so the MFMA work is 4 MACs per output and input channel
// instead of 9.  The eight-wave forms this kernel grew out of (round 1 / first half of round 2: one item per
// workgroup, pre-activated or GroupNorm-fused input) are kept for the record in scratch/wino_fused_8wave/.
// Result differs from the direct conv by re-association only.
#include "common.h"

#include <stdlib.h>

#include <type_traits>
#include <vector>


namespace kd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t OOB_OFF = 0x80000000u;   // buffer offset past num_records: the DMA writes zeros
constexpr int WF_K = 4;                     // input channels per chunk
constexpr int WF_RAW = 2048;                // 512 pixel slots x 4 floats (18 x 18 = 324 used)
constexpr int WF_UV = 16 * 64 * WF_K;       // one chunk of U (16 positions x 64 channels) or V (x 64 tiles)

// One chunk of U or V in LDS (and U in HBM, chunk after chunk): element (position p, row, k) with row = output
// channel (U) or tile (V) and k = channel within the chunk sits at [p/2][k/2][row][p%2][k%2].  A lane of the
// MFMA (row, k-half) then finds both k-steps of two positions in ONE 16-byte read, and the 32 rows of a
// k-half are contiguous: ds_read_b128 at the full LDS rate without bank conflicts (a [p][row][k] layout
// needs two 8-byte reads per pair, which hipcc merges into ds_read2st64_b64 at half that rate).
__host__ __device__ constexpr int64_t wf_uv_index(int64_t chunk, int p, int row, int k) {
  return chunk * WF_UV + ((((p >> 1) * 2 + (k >> 1)) * 64 + row) * 4 + (p & 1) * 2 + (k & 1));
}

// OIHW 3x3 weights -> U = G g G^T in the order the kernel's DMA reads: [N/64][C/4] chunks of wf_uv_index
__global__ __launch_bounds__(256) void wino_fused_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int N,
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
  const int nchunks = C / WF_K;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = r * 4 + s;
      U[wf_uv_index((int64_t)(n / 64) * nchunks + c / WF_K, p, n % 64, c % WF_K)] = u[s] * scale;
    }
  }
}

constexpr int WG16_MAXC = 2048;   // channels of the affine table kept in LDS

// ------------------------------------------------------------------------------------------------
// The same kernel as SIXTEEN waves (four per SIMD, 64 accumulator registers each, <= 128 VGPRs).  With two waves
// per SIMD the matrix pipe idles whenever both have a non-MFMA instruction at the head of their (in-order)
// streams - a DMA piece being issued, an LDS instruction waiting for its queue, an operand not back yet; removing
// the loop's barriers altogether changed nothing (KD_FWINO_VAR 4 / 5), so it is not the synchronisation.  Four
// waves make that coincidence rare.  Wave (pr, wm, wn) owns transformed ROW pr (positions 4 pr .. 4 pr + 3) of
// 32 tiles x 32 channels; thread = (tile, channel, row pr) of the input transform (8 patch reads, 8 VALU, 4 V
// stores); waves 0-7 (rows 0, 1) issue the raw-patch DMA, every wave one piece of U.  The output transform
// combines the four rows of a (wm, wn) tile through LDS: each wave finishes 4 of the 16 accumulator elements.
//
// PERSISTENT: the grid is one workgroup per CU and a workgroup walks the items (patch x 64-channel slab) id =
// blockIdx.x + k gridDim.x (gridDim.x % 8 == 0 keeps a workgroup's items on its XCD's share of the patch order).
// Behind the main loop's last barrier the raw-patch stages are free: the next item's first four raw chunks are
// fetched while this item's output transform, exchange and stores run; its U chunks 0 / 1 follow once the exchange
// (which lives in the U and V stages) has been read.  What a one-item workgroup pays per item - dispatch, descriptor
// set-up, the HBM latency of the first patch chunks - was 6.5 chunk times per item (17 % of a Cin = 128 layer).
template <bool STAMP>   // STAMP: diagnostic build that leaves s_memtime stamps of the item phases (KD_FWINO_STAMP)
__global__ __launch_bounds__(1024) void wino_fused_gn16_kernel(const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ ab,
                                                               const float* __restrict__ U,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ res, int ldres,
                                                               float* __restrict__ y, int B, int H, int W, int C,
                                                               int N, double* __restrict__ opart, int oG, const int4* __restrict__ items,
                                                               long long* __restrict__ stamps) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) float raw_0[WF_RAW], raw_1[WF_RAW], raw_2[WF_RAW], raw_3[WF_RAW];
  __shared__ __attribute__((aligned(1024))) float us_0[WF_UV], us_1[WF_UV], us_2[WF_UV], us_3[WF_UV];
  __shared__ __attribute__((aligned(1024))) float vs[2 * WF_UV];
  __shared__ __attribute__((aligned(16))) float abl[2 * WG16_MAXC];
  auto rawp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return raw_0; else if constexpr (decltype(S)::value == 1) return raw_1; else if constexpr (decltype(S)::value == 2) return raw_2; else return raw_3; };
  auto usp = [&](auto S) -> float* { if constexpr (decltype(S)::value == 0) return us_0; else if constexpr (decltype(S)::value == 1) return us_1; else if constexpr (decltype(S)::value == 2) return us_2; else return us_3; };

  // tq / lq: the thread and lq index as the loader, the activation and the transform see them.  They are re-made
  // opaque (empty asm) at the top of every item: whatever hipcc derives from them (slot coordinates, LDS addresses)
  // is then re-derived per item instead of being carried - as a loop invariant of the ITEM loop - through the
  // epilogue, where the registers are needed (it spilled 30 of them)
  int tq = threadIdx.x, lq = tq & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tq >> 6);
  const int pr = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  int stamp_item = 0;
  auto stamp = [&](int k) {   // [block][wave 0 | 15][item < 8][phase k < 8]
    if constexpr (STAMP) {
      if ((wave == 0 || wave == 15) && (threadIdx.x & 63) == 0 && stamp_item < 8)
        stamps[(((int64_t)blockIdx.x * 2 + (wave == 15)) * 8 + stamp_item) * 8 + k] = __builtin_amdgcn_s_memtime();
    }
  };
  const int pw = W / 16, ph_ = H / 16;
  const int nh = N / 64;
  const int npatch = B * pw * ph_;
  const int nitems = npatch * nh;
  const int nchunks = C / WF_K;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)((int64_t)16 * N * C * 4), 0x00020000);

  // ---- per-item state (the item being LOADED: during an item's epilogue this is already the next item)
  int b, prem, y0, x0, nhalf;
  uint32_t voffX;
  bool aok[2];
  __amdgpu_buffer_rsrc_t rsX;
  // items[id] = (image, y0, x0, 64-channel slab) of item id, in the order that keeps the N/64 items of one patch back
  // to back on ONE XCD (wino_fused_items_kernel): a table, because every one of the 16 waves decodes every item, and
  // the five divisions by run-time values cost each of them ~150 VALU + ~250 SALU instructions per item
  auto image_of = [&](int id) { return items[id].x; };
  auto slot_pixel = [&](int slot, int& iy, int& ix) {
    int py = slot / 18, pq = slot - py * 18;
    int px = pq < 9 ? 2 * pq : 2 * (pq - 9) + 1;
    iy = y0 - 1 + py;
    ix = x0 - 1 + px;
    return slot < 324 && iy >= 0 && iy < H && ix >= 0 && ix < W;
  };
  auto setup = [&](int id) {
    const int4 it = items[id];
    b = it.x;
    y0 = it.y;
    x0 = it.z;
    nhalf = it.w;
    prem = (y0 >> 4) * pw + (x0 >> 4);
    rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * H * W * ldx), 0, (int)((int64_t)H * W * ldx * 4),
                                            0x00020000);
    // (the empty asm makes the thread index opaque here: otherwise hipcc hoists the slot -> (row, column) divisions
    // of all three slots out of the item loop and carries them through the main loop in registers it does not have)
    int t_ = tq;
    asm volatile("" : "+v"(t_));
    int iy, ix;
    const bool ok = slot_pixel(t_, iy, ix);
    voffX = ok ? (uint32_t)(((iy * W + ix) * ldx) * 4) : OOB_OFF;
    // activation: values tq and tq + 1024 of the 324 x 4 patch floats; channel tq & 3, slots (tq >> 2) + 256 i
#pragma unroll
    for (int i = 0; i < 2; ++i) aok[i] = slot_pixel((t_ >> 2) + 256 * i, iy, ix);
  };
  auto issue_raw = [&](int chunk, auto S, auto LIVE) {   // waves 0-7 only: 512 pixel slots
    __attribute__((address_space(3))) float* rb = (__attribute__((address_space(3))) float*)(rawp(S) + wave * 256);
    const uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t)(chunk * WF_K * 4));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, rb, 16, (decltype(LIVE)::value || chunk < nchunks) ? voffX : OOB_OFF, sx,
                                             0, 0);
  };
  auto issue_u_at = [&](int chunk, auto S, auto LIVE, uint32_t toff) {   // every wave: 1 KB of the 16 KB chunk
    __attribute__((address_space(3))) float* ub = (__attribute__((address_space(3))) float*)(usp(S) + wave * 256);
    const uint32_t su = __builtin_amdgcn_readfirstlane((uint32_t)(((nhalf * nchunks + chunk) * WF_UV) * 4));
    const bool live = decltype(LIVE)::value || chunk < nchunks;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, ub, 16, live ? toff : OOB_OFF, su, 0, 0);
  };
  auto issue_u = [&](int chunk, auto S, auto LIVE) { issue_u_at(chunk, S, LIVE, (uint32_t)(tq * 16)); };
  auto activate = [&](int chunk, auto S, auto LIVE, auto MASK) {
    const int cc = (decltype(LIVE)::value ? chunk : min(chunk, nchunks - 1)) * WF_K + (tq & 3);
    const float2 a2 = *(const float2*)(abl + 2 * cc);
    float* ap = rawp(S) + tq;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < 1 || tq < 324 * 4 - 1024) {
        const float u = ap[i * 1024] * a2.x + a2.y;
        const float v = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
        ap[i * 1024] = (!decltype(MASK)::value || aok[i]) ? v : 0.f;
      }
    }
  };
  int tc, ttx, tty, tt;      // transform thread -> (channel, tile column, tile row, tile): set per item (derive)

  f32x16 acc[4];
  int aoff, boff;            // operand read offsets of this lane inside a V / U chunk: set per item (derive)
  auto derive = [&]() {
    tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    lq = tq & 63;
    const int t8 = tq & 255;
    tc = t8 & 3;
    ttx = (t8 >> 2) & 7;
    tty = t8 >> 5;
    tt = tty * 8 + ttx;
    const int frow = lq & 31, khalf = lq >> 5;
    aoff = (int)wf_uv_index(0, pr * 4, wm * 32 + frow, khalf * 2);
    boff = (int)wf_uv_index(0, pr * 4, wn * 32 + frow, khalf * 2);
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  using LiveT = std::integral_constant<bool, true>;
  using LiveF = std::integral_constant<bool, false>;
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  using R3 = std::integral_constant<int, 3>;
  auto run = [&](auto RR, auto MASK) {
    constexpr int R = decltype(RR)::value;           // transformed row of this wave's transform threads (= pr)
    constexpr bool LOADER = R < 2;                   // waves 0-7 carry the raw-patch DMA
    // row R of B^T d: d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3
    constexpr int IA = R == 0 ? 0 : R == 1 ? 1 : R == 2 ? 2 : 1;
    constexpr int IB = R == 0 ? 2 : R == 1 ? 2 : R == 2 ? 1 : 3;
    const int roff = (2 * tty * 18 + ttx) * 4 + tc;
    const int voff = (int)wf_uv_index(0, R * 4, tt, tc);
    constexpr int RS[4] = {0, 9, 1, 10};
    auto load_raw = [&](auto S, float (&e)[2][4]) {
      const float* rp = rawp(S) + roff;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        e[0][s] = rp[(IA * 18 + RS[s]) * 4];
        e[1][s] = rp[(IB * 18 + RS[s]) * 4];
      }
    };
    auto write_v = [&](int vstage, const float (&e)[2][4]) {
      float u[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) u[s] = R == 1 ? e[0][s] + e[1][s] : e[0][s] - e[1][s];
      float* v = vs + vstage * WF_UV + voff;
      v[0] = u[0] - u[2];
      v[2] = u[1] + u[2];
      v[2 * 64 * 4] = u[2] - u[1];
      v[2 * 64 * 4 + 2] = u[1] - u[3];
    };
    auto mfmas = [&](auto S, int vstage) {
      const float* va = vs + vstage * WF_UV + aoff;
      const float* ub = usp(S) + boff;
      float4 a4[2], b4[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a4[i] = *(const float4*)(va + i * 2 * 64 * 4);
        b4[i] = *(const float4*)(ub + i * 2 * 64 * 4);
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
    // iteration c (stages of chunk j: raw j % 4, U j % 4, V j % 2): raw(c+2) and U(c) have landed (issued two
    // iterations ago: a loader wave issues 2 pieces per iteration, the others 1), barrier, issue raw(c+4) and
    // U(c+2), MFMAs of chunk c, activate raw(c+2), transform raw(c+1) (activated one iteration ago) into V(c+1)
    auto body = [&](int c, auto Sc, auto Sc1, auto Sc2, auto LIVE) {
      if constexpr (LOADER) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (LOADER) issue_raw(c + 4, Sc, LIVE);
      issue_u(c + 2, Sc2, LIVE);
      constexpr int vcur = decltype(Sc)::value & 1;
      float e[2][4];
      load_raw(Sc1, e);
      mfmas(Sc, vcur);
      activate(c + 2, Sc2, LIVE, MASK);
      write_v(vcur ^ 1, e);
    };
    {
      float e[2][4];
      load_raw(S0{}, e);
      write_v(0, e);
    }
    int c = 0;
    for (; c + 8 <= nchunks; c += 4) {   // steady state: every chunk these four bodies prefetch exists
      body(c, S0{}, S1{}, S2{}, LiveT{});
      body(c + 1, S1{}, S2{}, S3{}, LiveT{});
      body(c + 2, S2{}, S3{}, S0{}, LiveT{});
      body(c + 3, S3{}, S0{}, S1{}, LiveT{});
    }
    for (; c < nchunks; c += 4) {
      body(c, S0{}, S1{}, S2{}, LiveF{});
      body(c + 1, S1{}, S2{}, S3{}, LiveF{});
      body(c + 2, S2{}, S3{}, S0{}, LiveF{});
      body(c + 3, S3{}, S0{}, S1{}, LiveF{});
    }
  };

  // exchange messages of the output transform: (group g = wm * 2 + wn, source row sr, destination index dd among the
  // other three rows) -> 2 KB (two float4 per lane); the 48 messages fill the V stages (16), U stages 2 and 3 (8
  // each) and the four raw stages (4 each).  U stages 0 / 1 stay free: the next item's U chunks 0 / 1 land there
  // while the exchange runs
  const int g = wm * 2 + wn;
  auto exmsg = [&](int sr, int dd) -> float4* {
    const int m = (g * 4 + sr) * 3 + dd;
    float* base = m < 16 ? vs + m * 512
                : m < 24 ? us_2 + (m - 16) * 512
                : m < 32 ? us_3 + (m - 24) * 512
                : m < 36 ? raw_0 + (m - 32) * 512
                : m < 40 ? raw_1 + (m - 36) * 512 : m < 44 ? raw_2 + (m - 40) * 512 : raw_3 + (m - 44) * 512;
    return (float4*)base;
  };

  int item = blockIdx.x;
  derive();
  setup(item);
  // first item: the classic prologue
  if (wave < 8) {
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
  float2 abv = make_float2(0.f, 0.f);
  if (tq < C) abv = ((const float2*)ab)[(int64_t)b * C + tq];
  bool first = true;
  while (true) {
    // ---- top of an item: its first raw chunks (and U chunks 0 / 1) are in flight or have landed
    stamp(0);
    if (!first) derive();
    if (first) {   // (later items: written behind the previous item's exchange, see there)
      if (tq < C) *(float2*)(abl + 2 * tq) = abv;
      if (tq + 1024 < C) *(float2*)(abl + 2 * (tq + 1024)) = ((const float2*)ab)[(int64_t)b * C + tq + 1024];
    }
    // first item: everything of the prologue.  Later items: the four 16-byte output stores of the previous item are
    // the youngest operations of every wave - all but 4 done means its prefetched raw 0-3, U 0 and U 1 have landed
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(1);
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
    const bool border = y0 == 0 || x0 == 0 || y0 + 16 >= H || x0 + 16 >= W;
    stamp(2);
    if (pr == 0) {
      if (border) run(R0{}, LiveT{}); else run(R0{}, LiveF{});
    } else if (pr == 1) {
      if (border) run(R1{}, LiveT{}); else run(R1{}, LiveF{});
    } else if (pr == 2) {
      if (border) run(R2{}, LiveT{}); else run(R2{}, LiveF{});
    } else {
      if (border) run(R3{}, LiveT{}); else run(R3{}, LiveF{});
    }
    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (out-of-range, zero) DMAs still write LDS
    __builtin_amdgcn_s_barrier();
    stamp(4);

    // ---- this item's coordinates for the epilogue; the loader state moves on to the next item, whose first four
    // raw chunks start now (the raw stages take no part in the exchange)
    const int eb = b, eprem = prem, ey0 = y0, ex0 = x0, en0 = nhalf * 64;
    int l_ = threadIdx.x & 63;   // opaque per item: the epilogue's addresses must not be hoisted over the main loop (registers)
    asm volatile("" : "+v"(l_));
    if (has_next && wave * 64 + l_ < C) abv = ((const float2*)ab)[(int64_t)image_of(next) * C + wave * 64 + l_];
    if (has_next) {   // U chunks 0 / 1 of the next item start now (their stages take no part in the exchange)
      setup(next);
      issue_u_at(0, S0{}, LiveF{}, (uint32_t)((wave * 64 + l_) * 16));   // (not the loop's tq * 16: see l_)
      issue_u_at(1, S1{}, LiveF{}, (uint32_t)((wave * 64 + l_) * 16));
    }

    // output transform Y = A^T m A.  Columns first, inside the wave: (q0, q1) = (a0 + a1 + a2, a1 - a2 - a3) of its
    // row.  Rows across the four waves of the (wm, wn) tile: Yrow0 = Q0 + Q1 + Q2, Yrow1 = Q1 - Q2 - Q3.  Wave pr
    // finishes the accumulator elements 4 pr .. 4 pr + 3 and hands the (q0, q1) of the other twelve to their owners
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
      stamp(5);
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
      // the exchange has been read by everybody: the next item's first four raw chunks may land on it (raw stages;
      // they arrive under the residual loads and the stores), and the V stages and U stages 2 / 3 become the waves'
      // private turn-around tiles below
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stamp(6);
      // the affine table of the next item's image: requested at the top of this epilogue, nobody reads the table
      // before the next main loop (a wait for it at the top of the item would also wait for this item's stores)
      if (has_next) {
        const int ti = wave * 64 + l_;
        if (ti < C) *(float2*)(abl + 2 * ti) = abv;
        if (ti + 1024 < C) *(float2*)(abl + 2 * (ti + 1024)) = ((const float2*)ab)[(int64_t)b * C + ti + 1024];
      }
      if (has_next && wave < 8) {
        issue_raw(0, S0{}, LiveF{});
        issue_raw(1, S1{}, LiveF{});
        issue_raw(2, S2{}, LiveF{});
        issue_raw(3, S3{}, LiveF{});
      }
      // The wave's 16 outputs per lane are tile row ty = wm * 4 + PR: pixel rows 2 ty, 2 ty + 1 x 16 pixels x 32
      // channels, one channel per lane.  Stored like that they are 16 dword stores (+ 16 residual loads) per wave, 512
      // narrow memory instructions per item whose ISSUE took ~6 chunk times per item, whatever the K (17 % of a
      // Cin = 128 layer; MI355X guide T21).  So the wave turns the block through a private 4 KB of LDS - [pixel p =
      // dy * 16 + x][channel], 16 ds_write_b32, 4 ds_read_b128, no barrier - and every lane finishes 4 consecutive
      // channels of 4 pixels with 16-byte accesses: 4 stores (+ 4 loads) per wave.
      float* sc = wave < 8 ? vs + wave * 1024 : wave < 12 ? us_2 + (wave - 8) * 1024 : us_3 + (wave - 12) * 1024;
      {
        float* sw = sc + (8 * (l_ >> 5)) * 32 + (l_ & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sw[(2 * i) * 32] = Q[0][i].x + Q[1][i].x + Q[2][i].x;
          sw[(2 * i + 1) * 32] = Q[0][i].y + Q[1][i].y + Q[2][i].y;
          sw[(16 + 2 * i) * 32] = Q[1][i].x - Q[2][i].x - Q[3][i].x;
          sw[(17 + 2 * i) * 32] = Q[1][i].y - Q[2][i].y - Q[3][i].y;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes have landed (nobody else reads them)
      const int rq = l_ >> 3, c4 = (l_ & 7) * 4;
      const int nb = en0 + wn * 32 + c4;
      const float4 b4v = bias ? *(const float4*)(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
      // per-image base pointers (scalar) + 32-bit offsets inside the image (H W N 4 < 2^31, host check)
      float* const yb = y + (int64_t)eb * H * W * N;
      const float* const rb = res + (int64_t)eb * H * W * ldres;
      const uint32_t pixr = (uint32_t)((ey0 + 2 * (wm * 4 + PR)) * W + ex0 + rq);
      float fs1 = 0.f, fs2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // pixel p = rq + 8 k of the block: row p >> 4 = k >> 1, x = rq + 8 (k & 1)
        const uint32_t pix = pixr + (uint32_t)((k >> 1) * W + 8 * (k & 1));
        float4 v = *(const float4*)(sc + (rq + 8 * k) * 32 + c4);
        v.x += b4v.x; v.y += b4v.y; v.z += b4v.z; v.w += b4v.w;
        if (res) {
          const float4 r4 = *(const float4*)(rb + (pix * (uint32_t)ldres + (uint32_t)nb));
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        *(float4*)(yb + (pix * (uint32_t)N + (uint32_t)nb)) = v;
        if (opart) {   // the lane's own 16 values in fp32 (31 operations at the fp32 rate), fp64 from there on
          fs1 += (v.x + v.y) + (v.z + v.w);
          fs2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fs2))));
        }
      }
      gs1 = (double)fs1;
      gs2 = (double)fs2;
    };
    if (pr == 0) finish(R0{});
    else if (pr == 1) finish(R1{});
    else if (pr == 2) finish(R2{});
    else finish(R3{});
    if (opart) {   // 8 entries per patch and 16-channel segment: (wm, pr)
      // a lane holds 4 channels 4 (l & 7) ..: lanes with (l & 7) < 4 make up segment 0, the others segment 1
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
      if ((l_ & ~4) == 0) {   // lanes 0 and 4: channel segments 0 and 1 of this wave
        int Cg = N / oG;
        asm volatile("" : "+s"(Cg));   // (opaque: its reciprocal is not to be kept in a register across the items)
        const int cabs = en0 + wn * 32 + 4 * (l_ & 4);
        const int gg = cabs / Cg, cseg = (cabs - gg * Cg) >> 4;
        const int npi = pw * ph_;
        const int64_t chunks = (int64_t)(Cg >> 4) * npi * 8;
        const int64_t entry = ((int64_t)cseg * npi + eprem) * 8 + (wm * 4 + pr);
        double* op = opart + (((int64_t)eb * oG + gg) * chunks + entry) * 2;
        op[0] = gs1;
        op[1] = gs2;
      }
    }
    stamp(7);
    ++stamp_item;
    if (!has_next) break;
    item = next;
    first = false;
  }
#endif
}

// the item table of the sixteen-wave kernel: id -> (image, y0, x0, slab)
__global__ __launch_bounds__(256) void wino_fused_items_kernel(int4* __restrict__ out, int B, int H, int W, int N) {
  const int pw = W / 16, ph = H / 16, nh = N / 64;
  const int npatch = B * pw * ph;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= npatch * nh) return;
  int bpatch, nhalf;
  if ((npatch & 7) == 0) {   // the N/64 items of one patch back to back on ONE XCD (ids go round-robin over the 8 XCDs)
    bpatch = (id / (8 * nh)) * 8 + (id & 7);
    nhalf = (id >> 3) % nh;
  } else {
    bpatch = id / nh;
    nhalf = id % nh;
  }
  const int b = bpatch / (pw * ph), prem = bpatch - b * pw * ph;
  out[id] = make_int4(b, (prem / pw) * 16, (prem % pw) * 16, nhalf);
}
size_t wino_fused_items_count(int B, int H, int W, int N) { return (size_t)B * (H / 16) * (W / 16) * (N / 64); }
int launch_wino_fused_items(void* items, int B, int H, int W, int N, hipStream_t s) {
  const size_t n = wino_fused_items_count(B, H, W, N);
  hipLaunchKernelGGL(wino_fused_items_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (int4*)items, B, H, W, N);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

bool wino_fused_ok(int B, int H, int W, int C, int N) {
  return B > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0 && C >= WF_K && C % WF_K == 0 && N >= 64 &&
         N % 64 == 0 && (int64_t)H * W * C * 4 < 0x7fffffff && (int64_t)16 * N * C * 4 < 0x7fffffff &&
         (int64_t)B * (H / 16) * (W / 16) * (N / 64) < 0x7fffffff;
}

int launch_wino_fused_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s, float scale) {
  KD_REQUIRE(O % 64 == 0 && I % WF_K == 0, "fused Winograd weights need Cout % 64 == 0 and Cin % 4 == 0");
  const int64_t total = (int64_t)O * I;
  hipLaunchKernelGGL(wino_fused_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, U, O, I,
                     scale);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ab[b][c] = (A, B) with GroupNorm(+FiLM)(x)[b][c] = A x + B (same arithmetic as gn_apply_silu_kernel)
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ scale_shift, int ld_ss,
                                                      float* __restrict__ ab, int B, int C, int G) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int b = idx / C, c = idx - b * C;
  const int g = c / (C / G);
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  float a = rstd * gamma[c];
  float bb = beta[c] - mean * a;
  if (scale_shift) {
    const float sc = scale_shift[(int64_t)b * ld_ss + c] + 1.0f;
    const float sh = scale_shift[(int64_t)b * ld_ss + C + c];
    a *= sc;
    bb = bb * sc + sh;
  }
  ab[2 * idx] = a * WF_AB_SCALE;   // the fused kernel's activation works on -log2(e) (A x + B)
  ab[2 * idx + 1] = bb * WF_AB_SCALE;
}

int launch_gn_fold(const float* stats, const float* gamma, const float* beta, const float* scale_shift, int ld_ss,
                   float* ab, int B, int C, int G, hipStream_t s) {
  KD_REQUIRE(C % G == 0, "gn_fold: C % G");
  hipLaunchKernelGGL(gn_fold_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, stats, gamma, beta, scale_shift, ld_ss, ab,
                     B, C, G);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

int wino_fused_gn_max_cin() { return WG16_MAXC; }

size_t wino_fused_out_stats_chunks(int H, int W, int N, int G) {
  return (size_t)(N / G / 16) * (H / 16) * (W / 16) * 8;
}

int launch_wino_fused_gn(const float* x, int ldx, const float* ab, const float* U, const float* bias, const float* res,
                         int ldres, float* y, int B, int H, int W, int C, int N, double* out_partial, int out_groups,
                         const void* items, hipStream_t s) {
  KD_REQUIRE(ldx >= C && ldx % 4 == 0 && (int64_t)H * W * ldx * 4 < 0x7fffffff && ((uintptr_t)x & 15) == 0,
             "GroupNorm-fused Winograd conv: bad input row stride");
  KD_REQUIRE(wino_fused_ok(B, H, W, C, N) && C <= WG16_MAXC,
             "GroupNorm-fused Winograd conv needs H, W % 16 == 0, Cin % 4 == 0, Cin <= 2048, Cout % 64 == 0");
  KD_REQUIRE(!out_partial || (out_groups > 0 && N % out_groups == 0 && (N / out_groups) % 16 == 0),
             "output statistics need groups of a multiple of 16 channels");
  const unsigned grid = (unsigned)((int64_t)B * (H / 16) * (W / 16) * (N / 64));
  {
    KD_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0 && ldres % 4 == 0,
               "GroupNorm-fused Winograd conv: output, bias and residual rows must be 16-byte aligned");
    KD_REQUIRE(items != nullptr, "GroupNorm-fused Winograd conv: item table missing (launch_wino_fused_items)");
    KD_REQUIRE((int64_t)H * W * N * 4 < 0x7fffffff && (!res || (int64_t)H * W * ldres * 4 < 0x7fffffff),
               "GroupNorm-fused Winograd conv: output / residual images above 2 GB");
    static int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      int dev = 0;
      KD_HIP_CHECK(hipGetDevice(&dev));
      KD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 8;   // a multiple of the 8 XCDs
    }
    const unsigned pgrid = grid < (unsigned)cus ? grid : (unsigned)cus;   // persistent: one workgroup per CU
#ifdef KD_FWINO_STAMP_BUILD   // diagnostic library only (make EXTRA=-DKD_FWINO_STAMP_BUILD): not in the product build
    static const bool stamping = kd_switch("KD_FWINO_STAMP", 0) != 0;   // (build with EXTRA="-DKD_FWINO_STAMP_BUILD -DKD_EXPERIMENT")
    if (stamping) {   // diagnostic: per-phase cycle counts of the first items of every workgroup, printed to stderr
      const size_t n = (size_t)pgrid * 2 * 8 * 8;
      long long* d = nullptr;
      KD_HIP_CHECK(hipMalloc((void**)&d, n * sizeof(long long)));
      KD_HIP_CHECK(hipMemsetAsync(d, 0, n * sizeof(long long), s));
      hipLaunchKernelGGL(wino_fused_gn16_kernel<true>, dim3(pgrid), dim3(1024), 0, s, x, ldx, ab, U, bias, res, ldres, y, B, H, W,
                         C, N, out_partial, out_groups, (const int4*)items, d);
      KD_HIP_CHECK(hipStreamSynchronize(s));
      std::vector<long long> h(n);
      KD_HIP_CHECK(hipMemcpy(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost));
      (void)hipFree(d);
      double acc[2][9] = {};
      long cnt[2] = {};
      for (unsigned blk = 0; blk < pgrid; ++blk)
        for (int w = 0; w < 2; ++w)
          for (int it = 1; it < 7; ++it) {   // items 1..6: steady state
            const long long* t = &h[(((size_t)blk * 2 + w) * 8 + it) * 8];
            const long long* tn = t + 8;
            if (!t[0] || !t[7] || !tn[0]) continue;
            for (int k = 0; k < 7; ++k) acc[w][k] += (double)(t[k + 1] - t[k]);
            acc[w][7] += (double)(tn[0] - t[7]);
            acc[w][8] += (double)(tn[0] - t[0]);
            ++cnt[w];
          }
      for (int w = 0; w < 2; ++w)
        if (cnt[w])
          fprintf(stderr, "fwino16 stamps C=%d N=%d H=%d wave %d (n=%ld): top-wait %.0f | activate %.0f | loop %.0f | drain %.0f | "
                          "q+exch-write %.0f | exch-read+B2 %.0f | turn+res+stores %.0f | to-next-top %.0f | item %.0f cycles\n",
                  C, N, H, w ? 15 : 0, cnt[w], acc[w][0] / cnt[w], acc[w][1] / cnt[w], acc[w][2] / cnt[w], acc[w][3] / cnt[w],
                  acc[w][4] / cnt[w], acc[w][5] / cnt[w], acc[w][6] / cnt[w], acc[w][7] / cnt[w], acc[w][8] / cnt[w]);
      return 0;
    }
#endif
    hipLaunchKernelGGL(wino_fused_gn16_kernel<false>, dim3(pgrid), dim3(1024), 0, s, x, ldx, ab, U, bias, res, ldres, y, B, H, W,
                       C, N, out_partial, out_groups, (const int4*)items, nullptr);
    KD_HIP_CHECK(hipGetLastError());
    return 0;
  }
}

}  // namespace kd
