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
#ifdef KD_EXPERIMENT   // ablations (timing only, wrong results): 1 no stores, 2 no MFMAs, 4 no vector work, 8 no DMA, 16 no operand reads
  const int flags = xflags;
#else
  constexpr int flags = 0;
#endif
  __shared__ __attribute__((aligned(1024))) float lds[L_END];
  using LP = __attribute__((address_space(3))) float*;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave >> 2, cb = wave & 3;   // (the four waves of a tile half sit on the four SIMDs)
  const int pw = W / 32, ph_ = H / 16;
  const int nitems = B * pw * ph_ * (N / F4_N);
  const int nchunks = C / F4_K;
  const __amdgpu_buffer_rsrc_t rsU =
      __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)((int64_t)36 * N * C * 4), 0x00020000);

  // ---- MFMA role: A = V[p][16 mb + (lane & 15)][lane >> 4], B = U[p][16 cb + (lane & 15)][lane >> 4]
  const int aoff = ((lane >> 4) * F4_TILES + mb * 16 + (lane & 15)) * 4;
  const int boff = ((lane >> 4) * F4_N + cb * 16 + (lane & 15)) * 4;

  // ---- transform role (waves 0-5): row i = wave of B^T d B for tile (ty, tx) and channels 2 chp, 2 chp + 1
  const int ttx = lane & 7, tchp = (lane >> 3) & 1, tty = lane >> 4;
  const int ttile = tty * 8 + ttx;
  // row i of B^T d = c0 d[r0] + c1 d[r1] + c2 d[r2] + c3 d[r3] (wave-uniform rows and coefficients)
  int r0, r1, r2, r3;
  float k0, k1, k2, k3;
  switch (wave) {
    case 0: r0 = 0; r1 = 2; r2 = 4; r3 = 0; k0 = 4.f; k1 = -5.f; k2 = 1.f; k3 = 0.f; break;
    case 1: r0 = 1; r1 = 2; r2 = 3; r3 = 4; k0 = -4.f; k1 = -4.f; k2 = 1.f; k3 = 1.f; break;
    case 2: r0 = 1; r1 = 2; r2 = 3; r3 = 4; k0 = 4.f; k1 = -4.f; k2 = -1.f; k3 = 1.f; break;
    case 3: r0 = 1; r1 = 2; r2 = 3; r3 = 4; k0 = -2.f; k1 = -1.f; k2 = 2.f; k3 = 1.f; break;
    case 4: r0 = 1; r1 = 2; r2 = 3; r3 = 4; k0 = 2.f; k1 = -1.f; k2 = -2.f; k3 = 1.f; break;
    default: r0 = 1; r1 = 3; r2 = 5; r3 = 1; k0 = 4.f; k1 = -5.f; k2 = 1.f; k3 = 0.f; break;
  }
  const int rbase = ((4 * tty) * F4_ROW + ttx) * 4 + tchp * 2;   // patch pixel (4 ty, 4 tx), channel pair
  const int ro0 = rbase + r0 * F4_ROW * 4, ro1 = rbase + r1 * F4_ROW * 4, ro2 = rbase + r2 * F4_ROW * 4,
            ro3 = rbase + r3 * F4_ROW * 4;
  // V store: positions 6 i .. 6 i + 5 of channel ch: [p/4][ch][tile][p%4]
  const int ti = wave < 6 ? wave : 0;
  const int p4a = (6 * ti) >> 2;   // group of the row's first position (slots 0-3 if i is even, 2-3 if odd)
  const int vwo = ((p4a * 4 + 2 * tchp) * F4_TILES + ttile) * 4;

  // ---- activation role: float2 element aoff0 (+ 128 r for waves 6-7) of the raw stage
  const int aoff0 = wave < 6 ? wave * 64 + lane : 384 + (wave - 6) * 64 + lane;
  const int nact = wave < 6 ? 1 : 7;

  // ---- per-item state
  int b, y0, x0, slab, prem;
  __amdgpu_buffer_rsrc_t rsX;
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
    rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * H * W * ldx), 0, (int)((int64_t)H * W * ldx * 4),
                                            0x00020000);
    int iy, ix;
    voffA = voffB = F4_OOB;
    if (rawp0 >= 0 && slot_pixel(rawp0 * 64 + lane, iy, ix)) voffA = (uint32_t)(((iy * W + ix) * ldx) * 4);
    if (rawp1 >= 0 && slot_pixel(rawp1 * 64 + lane, iy, ix)) voffB = (uint32_t)(((iy * W + ix) * ldx) * 4);
    amask = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r)
      if (r < nact && slot_pixel((aoff0 + 128 * r) >> 1, iy, ix)) amask |= 1u << r;
  };
  auto issue_raw = [&](int chunk, int stage_off) {
    const uint32_t sx = __builtin_amdgcn_readfirstlane((uint32_t)(chunk * F4_K * 4));
    if (rawp0 >= 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (LP)(lds + stage_off + rawp0 * 256), 16, voffA, sx, 0, 0);
    if (rawp1 >= 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (LP)(lds + stage_off + rawp1 * 256), 16, voffB, sx, 0, 0);
  };
  // 36 pieces of 1 KB: wave w carries 4 w .. 4 w + 3 and (w < 4) 32 + w; part 0 = the first two, part 1 = the rest
  auto issue_u_part = [&](int chunk, int stage_off, int part) {
    const uint32_t su = __builtin_amdgcn_readfirstlane((uint32_t)(((slab * nchunks + chunk) * F4_U) * 4));
    const uint32_t vo = (uint32_t)(lane * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((q >> 1) == part)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (LP)(lds + stage_off + (wave * 4 + q) * 256), 16, vo,
                                                 su + (uint32_t)((wave * 4 + q) * 1024), 0, 0);
    if (wave < 4 && part == 1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (LP)(lds + stage_off + (32 + wave) * 256), 16, vo,
                                               su + (uint32_t)((32 + wave) * 1024), 0, 0);
  };
  auto issue_u = [&](int chunk, int stage_off) {
    issue_u_part(chunk, stage_off, 0);
    issue_u_part(chunk, stage_off, 1);
  };
  // GroupNorm / FiLM / SiLU of one raw stage in place: 1224 float2 elements (612 pixel slots x 2 channel pairs) per chunk.
  // Waves 0-5 take one round of 64 elements each behind their transform, waves 6-7 seven rounds each (the transform is
  // worth five rounds: the two waves of a SIMD, w and w + 4, carry about the same vector work).  ab holds -log2(e) (A, B):
  // u = -log2(e) v, e^-v = 2^u, u / (1 + 2^u) = -log2(e) SiLU(v); -ln 2 sits in U (WF_U_SCALE).
  // All reads first, then the arithmetic, then the writes: one LDS round trip per chunk, not one per round
  auto activate = [&](int chunk, int stage_off, auto MASK) {
    const f32x4 a4 = *(const f32x4*)(lds + L_AB + 2 * (chunk * F4_K + (lane & 1) * 2));
    f32x2* ap = (f32x2*)(lds + stage_off + aoff0 * 2);
    if (wave < 6) {
      f32x2 v = ap[0];
      const float u0 = fmaf(v[0], a4[0], a4[1]), u1 = fmaf(v[1], a4[2], a4[3]);
      v[0] = u0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u0));
      v[1] = u1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u1));
      if (decltype(MASK)::value && !(amask & 1)) v[0] = v[1] = 0.f;
      ap[0] = v;
    } else {
      f32x2 v[7];
#pragma unroll
      for (int r = 0; r < 7; ++r)
        if (r < 6 || aoff0 + 128 * 6 < F4_SLOTS * 2) v[r] = ap[128 * r];   // (wave 7's last round is partly past the 612 slots)
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        const float u0 = fmaf(v[r][0], a4[0], a4[1]), u1 = fmaf(v[r][1], a4[2], a4[3]);
        v[r][0] = u0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u0));
        v[r][1] = u1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u1));
        if (decltype(MASK)::value && !((amask >> r) & 1)) v[r][0] = v[r][1] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 7; ++r)
        if (r < 6 || aoff0 + 128 * 6 < F4_SLOTS * 2) ap[128 * r] = v[r];
    }
  };
  // row i of B^T d B for (tile, 2 channels) from the activated raw stage into the V stage (waves 0-5)
  auto transform = [&](int raw_off, int v_off) {
    const float* R = lds + raw_off;
    // column c of the tile: pixel 4 tx + c -> slot offset {0, 9, 18, 26, 1, 10} (residue block, then index)
    constexpr int CO[6] = {0, 9 * 4, 18 * 4, 26 * 4, 1 * 4, 10 * 4};
    float wa[6], wb[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const f32x2 d0 = *(const f32x2*)(R + ro0 + CO[c]), d1 = *(const f32x2*)(R + ro1 + CO[c]),
                  d2 = *(const f32x2*)(R + ro2 + CO[c]), d3 = *(const f32x2*)(R + ro3 + CO[c]);
      wa[c] = fmaf(k3, d3[0], fmaf(k2, d2[0], fmaf(k1, d1[0], k0 * d0[0])));
      wb[c] = fmaf(k3, d3[1], fmaf(k2, d2[1], fmaf(k1, d1[1], k0 * d0[1])));
    }
    float ta[6], tb[6];
    f4_bt(wa, ta);
    f4_bt(wb, tb);
    float* Vd = lds + v_off + vwo;
    constexpr int CH = F4_TILES * 4;   // channel stride
    constexpr int PG = 4 * F4_TILES * 4;   // position-group stride
    if ((wave & 1) == 0) {   // positions 6 i .. 6 i + 3 fill one group, 6 i + 4, 6 i + 5 open the next
      *(f32x4*)(Vd) = f32x4{ta[0], ta[1], ta[2], ta[3]};
      *(f32x4*)(Vd + CH) = f32x4{tb[0], tb[1], tb[2], tb[3]};
      *(f32x2*)(Vd + PG) = f32x2{ta[4], ta[5]};
      *(f32x2*)(Vd + PG + CH) = f32x2{tb[4], tb[5]};
    } else {                 // positions 6 i, 6 i + 1 close a group (slots 2, 3), 6 i + 2 .. 6 i + 5 fill the next
      *(f32x2*)(Vd + 2) = f32x2{ta[0], ta[1]};
      *(f32x2*)(Vd + CH + 2) = f32x2{tb[0], tb[1]};
      *(f32x4*)(Vd + PG) = f32x4{ta[2], ta[3], ta[4], ta[5]};
      *(f32x4*)(Vd + PG + CH) = f32x4{tb[2], tb[3], tb[4], tb[5]};
    }
  };

  f32x4 acc[36];
  auto mfmas = [&](int u_off, int v_off, auto between) {
    const float* va = lds + v_off + aoff;
    const float* ub = lds + u_off + boff;
    // three batches of three position groups; the reads of batch k + 1 are in flight under the MFMAs of batch k
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
    between(0);
    mm(3, a1, b1);
    between(1);
    mm(6, a0, b0);
  };

  constexpr int RS[3] = {L_R0, L_R1, L_R2};
  constexpr int US[2] = {L_U0, L_U1};
  constexpr int VS[2] = {L_V0, L_V1};
  (void)US[1];
  (void)VS[1];

  using MaskT = std::integral_constant<bool, true>;
  using MaskF = std::integral_constant<bool, false>;
  bool border = false;
  // the vector work of one interval: transform of chunk ct (waves 0-5), activation of chunk ca (all waves)
  auto vector_work = [&](int ct, int raw_t, int v_t, int ca, int raw_a) {
    if (wave < 6 && ct < nchunks && !(flags & 64)) transform(raw_t, v_t);
    if (ca < nchunks && !(flags & 32)) {
      if (border) activate(ca, raw_a, MaskT{});
      else activate(ca, raw_a, MaskF{});
    }
  };
  // first raw chunks of the item that setup() just described (the three raw stages are free)
  auto issue_first_raws = [&]() {
    issue_raw(0, RS[0]);
    if (1 < nchunks) issue_raw(1, RS[1]);
    if (2 < nchunks) issue_raw(2, RS[2]);
  };

  int item = blockIdx.x;
  setup(item);
  issue_first_raws();
  f32x2 abv = f32x2{0.f, 0.f};
  if (tid < C) abv = ((const f32x2*)ab)[(int64_t)b * C + tid];
  bool first = true;
  while (true) {
    // ---- top of an item: its raw chunks 0-2 are in flight (issued under the previous item's epilogue), abv holds its
    // affine table row.  (Later items: the only younger operations of a wave are its last 8 output stores.)
    if (tid < C) *(f32x2*)(lds + L_AB + 2 * tid) = abv;
    border = y0 == 0 || x0 == 0 || y0 + 16 >= H || x0 + 32 >= W;
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_u(0, US[0]);
    vector_work(nchunks, 0, 0, 0, RS[0]);           // activate chunk 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vector_work(0, RS[0], VS[0], 1, RS[1]);         // transform chunk 0, activate chunk 1

    // stage offsets of chunk c: raw c % 3, U / V c % 2 (scalars, rotated by hand)
    int r_c = L_R0, r_c1 = L_R1, r_c2 = L_R2;   // raw stages of chunks c, c + 1, c + 2
    int u_c = L_U0, u_c1 = L_U1, v_c = L_V0, v_c1 = L_V1;
    for (int c = 0; c < nchunks; ++c) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // The two waves of a SIMD (w and w + 4) run the interval in opposite order - matrix work first on waves 0-3, vector
      // work first on waves 4-7 - so that one's transform / activation issues under the other's MFMAs
      // (ONE copy of the MFMA block: two copies behind a branch make hipcc spill the 144 accumulators).
      // The DMA of the next chunks (6 one-KB pieces per wave: raw(c+3) into the stage the transform of iteration c - 1
      // consumed, U(c+1)) costs the issuing wave ~100 cycles a piece: waves 4-7 issue theirs in front of their vector
      // work, waves 0-3 between the batches of their MFMAs, where the partner's MFMAs cover it
      const bool dma = !(flags & 8);
      if (wave >= 4) {
        if (c + 3 < nchunks && dma) issue_raw(c + 3, r_c);
        if (c + 1 < nchunks && dma) issue_u(c + 1, u_c1);
        if (!(flags & 4)) vector_work(c + 1, r_c1, v_c1, c + 2, r_c2);
      }
      if (!(flags & 2))
        mfmas(u_c, v_c, [&](int k) {
          if (wave < 4 && dma) {
            if (k == 0) {
              if (c + 3 < nchunks) issue_raw(c + 3, r_c);
              if (c + 1 < nchunks) issue_u_part(c + 1, u_c1, 0);
            } else if (c + 1 < nchunks) {
              issue_u_part(c + 1, u_c1, 1);
            }
          }
        });
      if (wave < 4 && !(flags & 4)) vector_work(c + 1, r_c1, v_c1, c + 2, r_c2);
      const int t3 = r_c;
      r_c = r_c1;
      r_c1 = r_c2;
      r_c2 = t3;
      const int tu = u_c;
      u_c = u_c1;
      u_c1 = tu;
      const int tv = v_c;
      v_c = v_c1;
      v_c1 = tv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every stage is free from here on

    // ---- this item's coordinates for the epilogue; the loader state moves on to the next item, whose first three raw
    // chunks start now (the raw stages take no part in the output exchange) and whose affine row is fetched
    const int eb = b, ey0 = y0, ex0 = x0, eslab = slab, eprem = prem;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;
    if (has_next) {
      setup(next);
      issue_first_raws();
      if (tid < C) abv = ((const f32x2*)ab)[(int64_t)b * C + tid];
    }

    // ---- output transform Y = A^T m A in registers (the lane holds positions 0..35 of tiles 16 mb + 4 (lane >> 4) + r at
    // output channel 16 cb + (lane & 15)), then a turn through LDS so that every pixel's 64 channels (256 bytes) leave
    // in 16-byte pieces: X[tile 0..15][pixel 0..15][64 channels] = 64 KB over the U stages, one half of the tiles (mb)
    // per round; all eight waves read a round back (32 pixels each), add bias / residual and store
    float* const X = lds + L_U0;
    const int n4 = eslab * F4_N + (lane & 15) * 4;
    const f32x4 b4 = bias ? *(const f32x4*)(bias + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* const yb = y + (int64_t)eb * H * W * N;
    const float* const rb = res ? res + (int64_t)eb * H * W * ldres : nullptr;
    float fs1 = 0.f, fs2 = 0.f;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      if (mb == h) {
        float* xw = X + ((lane >> 4) * 4 * 16) * F4_N + cb * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
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
            for (int q = 0; q < 4; ++q) xw[((r * 16) + a * 4 + q) * F4_N] = o[q];
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int P = wave * 32 + j * 4 + (lane >> 4);   // pixel of the round: tile P >> 4, row (P >> 2) & 3, column P & 3
        const int t = h * 16 + (P >> 4);
        const uint32_t pix = (uint32_t)((ey0 + 4 * (t >> 3) + ((P >> 2) & 3)) * W + ex0 + 4 * (t & 7) + (P & 3));
        f32x4 v = *(const f32x4*)(X + P * F4_N + (lane & 15) * 4);
        v += b4;
        if (rb && !(flags & 1)) v += *(const f32x4*)(rb + (pix * (uint32_t)ldres + (uint32_t)n4));
        if (!(flags & 1) || v[0] == 123.456f) *(f32x4*)(yb + (pix * (uint32_t)N + (uint32_t)n4)) = v;
        fs1 += (v[0] + v[1]) + (v[2] + v[3]);
        fs2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], fs2))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // the round has been read: the next round (or the next item's U) may overwrite it
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
