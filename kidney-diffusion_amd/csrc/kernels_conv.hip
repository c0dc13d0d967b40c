// Implicit-GEMM convolution / token GEMM on the fp32 matrix cores of gfx950.
//
//   y[m][n] = epilogue( sum_{tap} sum_{c} x[pixel(m, tap)][c] * w[tap][n][c] )
//
// m runs over the B*Ho*Wo output pixels (NHWC order), n over Cout.  Both operands are staged
// into LDS as rows of BK = 32 contiguous k (channels of one tap), so the A tile (activations)
// and the B tile (weights, packed [tap][Cout][Cin] at plan-build time) share one loader.
// The product runs on v_mfma_f32_32x32x2_f32: exact fp32 (bitwise a k-ordered fmaf chain),
// 64 FLOP/clk/SIMD, which is the fp32 roofline of the chip (157 TFLOP/s).
//
// LDS rows are padded to 36 floats: a ds_read_b128 of 16 lanes on 16 different rows then
// touches 16 different 16-B slots of the 256-B bank row (conflict-free).
// Each lane's b128 read supplies 4 consecutive k for 4 MFMA steps: lanes 0-31 (k-half 0) read
// k = 8j..8j+3 and lanes 32-63 read k = 8j+4..8j+7; A and B use the same map, and k is only a
// summation label, so the pairing is consistent.
#include "common.h"
#include "epilogue.h"

#include <stdlib.h>

namespace kd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + __expf(-v));
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
  return v;
}

// Finishes 4 consecutive output channels n..n+3 of output pixel m (vec) or one channel (scalar).
template <int W>
__device__ __forceinline__ void epilogue_store(const ConvParams& p, int64_t m, int n, float* v) {
  const int hw_o = p.Ho * p.Wo;
  const int b = (int)(m / hw_o);
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = act_apply(v[e] + (p.bias ? p.bias[n + e] : 0.f), p.act);
  if (p.gate_src) {
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] += p.gate_src[m * p.ldgs + n + e] * p.gate[(int64_t)b * p.Cout + n + e];
  }
  if (p.res) {
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] += p.res[m * p.ldres + n + e];
  }
  float* dst;
  if (p.out_mode == OUT_NHWC) {
    dst = p.y + m * p.ldy + p.yoff + n;
  } else if (p.out_mode == OUT_PIXSHUF) {
    // weight rows were packed as n' = (i*2+j)*Co + c ; output NHWC [B][2Ho][2Wo][Co]
    int Co = p.Cout >> 2;
    int q = n / Co, c = n - q * Co;
    int rem = (int)(m - (int64_t)b * hw_o);
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    int64_t o = (((int64_t)b * 2 * p.Ho + 2 * oy + (q >> 1)) * (2 * p.Wo) + 2 * ox + (q & 1));
    dst = p.y + o * p.ldy + p.yoff + c;
  } else {  // OUT_NCHW planar [B][Cout][Ho][Wo] (scalar path only)
    int rem = (int)(m - (int64_t)b * hw_o);
    dst = p.y + ((int64_t)b * p.Cout + n) * hw_o + rem;
  }
  if (W == 4) {
    f32x4 o4 = {v[0], v[1], v[2], v[3]};
    *(f32x4*)dst = o4;
  } else {
    dst[0] = v[0];
  }
}

// ---- shared epilogue.  The accumulators go through LDS so that global stores (and the residual /
// gate reads) are 16 B per lane along n.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Callers end their main loop on a barrier: LDS is free.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void store_tile(const ConvParams& p, float* lds,
                                           f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], int64_t m0, int n0,
                                           int64_t M) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int C_LD = WAVES_N * 32 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const bool vec_ok = p.out_mode != OUT_NCHW && (p.Cout & 3) == 0 && (p.ldy & 3) == 0 && (p.yoff & 3) == 0 &&
                      (((uintptr_t)p.y) & 15) == 0 && (!p.res || ((p.ldres & 3) == 0 && (((uintptr_t)p.res) & 15) == 0)) &&
                      (!p.gate_src || (p.ldgs & 3) == 0) && (p.out_mode != OUT_PIXSHUF || (p.Cout & 15) == 0);
  float* Cs = lds;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rowb = (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(rowb + (r & 3) + 8 * (r >> 2)) * C_LD + wn * 32 + (lane & 31)] = acc[i][j][r];
    }
    __syncthreads();
    constexpr int V_PER_ROW = WAVES_N * 8;
    if (vec_ok) {
      for (int idx = tid; idx < BM * V_PER_ROW; idx += NT) {
        int row = idx / V_PER_ROW, c4 = idx - row * V_PER_ROW;
        int wn_ = c4 >> 3, c = (c4 & 7) * 4;
        int n = n0 + (wn_ * TN + j) * 32 + c;
        int64_t m = m0 + row;
        if (m < M && n < p.Cout) {
          f32x4 t = *(const f32x4*)(Cs + row * C_LD + wn_ * 32 + c);
          float v[4] = {t[0], t[1], t[2], t[3]};
          epilogue_store<4>(p, m, n, v);
        }
      }
    } else {
      for (int idx = tid; idx < BM * WAVES_N * 32; idx += NT) {
        int row = idx / (WAVES_N * 32), cc = idx - row * (WAVES_N * 32);
        int wn_ = cc >> 5, c = cc & 31;
        int n = n0 + (wn_ * TN + j) * 32 + c;
        int64_t m = m0 + row;
        if (m < M && n < p.Cout) {
          float v[1] = {Cs[row * C_LD + wn_ * 32 + c]};
          epilogue_store<1>(p, m, n, v);
        }
      }
    }
    __syncthreads();
  }
}

// ---- NHWC epilogue straight from the accumulator registers (fast path).  In the C/D map a lane owns
// ONE output channel (col = lane&31) of 16 rows, and a half-wave covers 32 consecutive channels of a
// row: every global store / residual / gate read is a full 128-B line, with no LDS round trip and no
// barriers (measured against store_tile on the Winograd GEMMs: +5..9 %).  bias and the per-image gate
// are one scalar per lane.  Needs hw_o % 32 == 0 when a gate is applied (one image per 32-row MFMA tile).
// GroupNorm partials of the stored tile for the layer that reads the map (ConvParams::seg_partial, SegSrc in
// common.h): (s1, s2) = this lane's sum / sum of squares over its 16 rows of one channel; the 16 lanes x 2
// half-waves of a 16-channel segment are folded with shuffles and lanes 0 / 16 write the segment's partial of
// this 32-row tile (chunk).  All 64 lanes call it (shuffles); `valid` masks columns past Cout.
__device__ __forceinline__ void seg_partial_store(double* __restrict__ seg, int nseg, int64_t nchunk, int64_t b,
                                                  int64_t chunk, int ch, bool valid, double s1, double s2) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off <= 8; off <<= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if ((lane & 47) == 0 && valid) {   // lanes 0 and 16
    double* o = seg + ((b * nseg + (ch >> 4)) * nchunk + chunk) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool GATE, bool RES, bool ACT, bool STATS>
__device__ __forceinline__ void store_tile_regs_impl(const ConvParams& p,
                                                     f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                     int64_t m0, int n0, int64_t M) {
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int hw_o = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool n_ok = n < p.Cout;
    if (!STATS && !n_ok) continue;
    if (STATS && n0 + (wn * TN + j) * 32 >= p.Cout) continue;   // wave-uniform: the whole column block is outside
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t mt = m0 + (wm * TM + i) * 32;  // first row of this 32x32 MFMA tile
      if (mt >= M) continue;
      float gate = 0.f;
      if (GATE && n_ok) gate = p.gate[(mt / hw_o) * p.Cout + n];
      const int64_t mb = mt + 4 * (lane >> 5);
      float* yp = p.y + mb * p.ldy + p.yoff + n;
      const float* gp = GATE ? p.gate_src + mb * p.ldgs + n : nullptr;
      const float* rp = RES ? p.res + mb * p.ldres + n : nullptr;
      const bool full = mt + 32 <= M;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (n_ok && (full || mb + dr < M)) {
          float v = acc[i][j][r] + bias;
          if (ACT) v = act_apply(v, p.act);
          if (GATE) v += gp[(int64_t)dr * p.ldgs] * gate;
          if (RES) v += rp[(int64_t)dr * p.ldres];
          yp[(int64_t)dr * p.ldy] = v;
          if (STATS) {
            s1 += (double)v;
            s2 += (double)v * (double)v;
          }
        }
      }
      if (STATS) {
        const int64_t b = mt / hw_o;
        seg_partial_store(p.seg_partial, p.seg_nseg, hw_o >> 5, b, (mt - b * hw_o) >> 5, p.yoff + n - p.seg_c0, n_ok, s1, s2);
      }
    }
  }
}

// PixelShuffle(2) epilogue straight from the registers (upsample convs: weight rows packed n' = (i*2+j)*Co + c).
// Needs Co % 32 == 0 (the 32 channels of a lane group share (i, j) and stay one 128-B line) and Wo % 32 == 0
// (the 32 rows of an MFMA tile lie in one image row: one division per tile instead of one per element).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool STATS>
__device__ __forceinline__ void store_tile_regs_pixshuf(const ConvParams& p,
                                                        f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                        int64_t m0, int n0, int64_t M) {
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int Co = p.Cout >> 2;
  const int hw_o = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool n_ok = n < p.Cout;
    if (!STATS && !n_ok) continue;
    if (STATS && n0 + (wn * TN + j) * 32 >= p.Cout) continue;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    const int q = n / Co, c = n - q * Co;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t mt = m0 + (wm * TM + i) * 32;
      if (mt >= M) continue;
      const int64_t row = mt / p.Wo;                 // b * Ho + oy: the whole tile is in this image row
      const int ox0 = (int)(mt - row * p.Wo) + 4 * (lane >> 5);
      float* yp = p.y + ((2 * row + (q >> 1)) * (2 * p.Wo) + 2 * ox0 + (q & 1)) * (int64_t)p.ldy + p.yoff + c;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bias;
        if (p.act != ACT_NONE) v = act_apply(v, p.act);
        if (n_ok) yp[(int64_t)(2 * dr) * p.ldy] = v;
        if (STATS && n_ok) {
          s1 += (double)v;
          s2 += (double)v * (double)v;
        }
      }
      if (STATS) {   // chunk = (input row tile) * 4 + sub-position: each 32-row tile feeds 4 output positions
        const int64_t b = mt / hw_o;
        seg_partial_store(p.seg_partial, p.seg_nseg, (int64_t)(hw_o >> 5) * 4, b, ((mt - b * hw_o) >> 5) * 4 + q,
                          p.yoff + c - p.seg_c0, n_ok, s1, s2);
      }
    }
  }
}

// The same epilogue with 16-byte accesses (epilogue.h): each 32 x 32 accumulator tile goes through 4 KB of LDS private
// to the wave and leaves as 4 stores (+ 4 loads per added map) instead of 16 (+ 16).  A workgroup's vector-memory
// instructions - DMA pieces, epilogue loads and stores - share one address pipe per CU, and the short-K 1x1 convs
// on the big maps are bound by it: a 128 x 64 x 256 tile is 48 DMA pieces and 64 four-byte epilogue instructions
// per wave.  Needs full 32-row tiles inside one image, Cout % 4 == 0 and 16-byte aligned rows (conv_wide_ok).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool STATS>
__device__ __forceinline__ void store_tile_regs_wide(const ConvParams& p, float* lds,
                                                     f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], int64_t m0,
                                                     int n0, int64_t M) {
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int hw_o = p.Ho * p.Wo;
  float* scratch = lds + wave * 1024;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nc = n0 + (wn * TN + j) * 32;   // first column of the tile
    if (nc >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t mt = m0 + (wm * TM + i) * 32;
      if (mt >= M) continue;
      WideEpilogue e;
      e.y = p.y + mt * p.ldy + p.yoff + nc;
      e.ldy = p.ldy;
      e.bias = p.bias ? p.bias + nc : nullptr;
      e.res = p.res ? p.res + mt * p.ldres + nc : nullptr;
      e.ldres = p.ldres;
      e.gate_src = p.gate_src ? p.gate_src + mt * p.ldgs + nc : nullptr;
      e.ldgs = p.ldgs;
      e.gate = p.gate_src ? p.gate + (mt / hw_o) * p.Cout + nc : nullptr;
      e.rows = M - mt < 32 ? (int)(M - mt) : 32;
      e.cols = p.Cout - nc < 32 ? p.Cout - nc : 32;
      e.act = p.act;
      double s1, s2;
      store_tile32_wide<STATS>(*(const ep_f32x16*)&acc[i][j], scratch, e, s1, s2);
      if (STATS) {
        reduce_tile32_stats(s1, s2);
        if ((lane & ~4) == 0 && 4 * (lane & 7) < e.cols) {   // lanes 0 and 4: columns 0-15 and 16-31 of the tile
          const int64_t b = mt / hw_o;
          const int ch = p.yoff + nc + 4 * (lane & 4) - p.seg_c0;
          double* o = p.seg_partial + ((b * p.seg_nseg + (ch >> 4)) * (hw_o >> 5) + ((mt - b * hw_o) >> 5)) * 2;
          o[0] = s1;
          o[1] = s2;
        }
      }
    }
  }
}

// PixelShuffle(2) form of the 16-byte epilogue: the 32 rows of a tile are 32 consecutive input pixels of one image row
// (Wo % 32 == 0) and its 32 columns one sub-position q = (i, j) x 32 consecutive channels (Co % 32 == 0), so the
// tile is 32 output pixels two apart: the same turn with a row stride of 2 ldy.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool STATS>
__device__ __forceinline__ void store_tile_regs_pixshuf_wide(const ConvParams& p, float* lds,
                                                             f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                             int64_t m0, int n0, int64_t M) {
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int Co = p.Cout >> 2;
  const int hw_o = p.Ho * p.Wo;
  float* scratch = lds + wave * 1024;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nc = n0 + (wn * TN + j) * 32;
    if (nc >= p.Cout) continue;
    const int q = nc / Co, c = nc - q * Co;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t mt = m0 + (wm * TM + i) * 32;
      if (mt >= M) continue;
      const int64_t row = mt / p.Wo;                 // b * Ho + oy: the whole tile is in this image row
      const int ox0 = (int)(mt - row * p.Wo);
      WideEpilogue e;
      e.y = p.y + ((2 * row + (q >> 1)) * (2 * p.Wo) + 2 * ox0 + (q & 1)) * (int64_t)p.ldy + p.yoff + c;
      e.ldy = 2 * (int64_t)p.ldy;
      e.bias = p.bias ? p.bias + nc : nullptr;
      e.res = nullptr;
      e.ldres = 0;
      e.gate_src = nullptr;
      e.ldgs = 0;
      e.gate = nullptr;
      e.rows = 32;
      e.cols = 32;
      e.act = p.act;
      double s1, s2;
      store_tile32_wide<STATS>(*(const ep_f32x16*)&acc[i][j], scratch, e, s1, s2);
      if (STATS) {   // chunk = (input row tile) * 4 + sub-position
        reduce_tile32_stats(s1, s2);
        if ((lane & ~4) == 0) {
          const int64_t b = mt / hw_o;
          const int ch = p.yoff + c + 4 * (lane & 4) - p.seg_c0;
          double* o = p.seg_partial + ((b * p.seg_nseg + (ch >> 4)) * ((int64_t)(hw_o >> 5) * 4) + ((mt - b * hw_o) >> 5) * 4 + q) * 2;
          o[0] = s1;
          o[1] = s2;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void store_tile_regs(const ConvParams& p,
                                                f32x16 (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32], int64_t m0,
                                                int n0, int64_t M) {
  // the (wave-uniform) epilogue variant is chosen once, outside the unrolled store loops
  const bool act = p.act != ACT_NONE;
  if (p.seg_partial) {   // output feeds a GroupNorm: the variants that occur in the plans
    if (p.gate_src)
      store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, true, false, false, true>(p, acc, m0, n0, M);
    else if (p.res)
      store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, true, false, true>(p, acc, m0, n0, M);
    else
      store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, false, false, true>(p, acc, m0, n0, M);
    return;
  }
  if (p.gate_src) {
    store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, true, false, false, false>(p, acc, m0, n0, M);  // resblock tail
  } else if (p.res) {
    if (act)
      store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, true, true, false>(p, acc, m0, n0, M);
    else
      store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, true, false, false>(p, acc, m0, n0, M);
  } else if (act) {
    store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, false, true, false>(p, acc, m0, n0, M);
  } else {
    store_tile_regs_impl<BM, BN, WAVES_M, WAVES_N, false, false, false, false>(p, acc, m0, n0, M);
  }
}

constexpr int BK = 32;
constexpr int LDS_LD = 36;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_igemm_kernel(ConvParams p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int ROWS_PER_PASS = NT / 8;
  constexpr int A_PASSES = BM / ROWS_PER_PASS;
  constexpr int B_PASSES = BN / ROWS_PER_PASS;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/loader mismatch");

  constexpr int C_LD = WAVES_N * 32 + 4;  // epilogue staging: one 32-column group per wave column
  constexpr int LDS_FLOATS = (BM + BN) * LDS_LD > BM * C_LD ? (BM + BN) * LDS_LD : BM * C_LD;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* As = lds;
  float* Bs = lds + BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- loader coordinates
  const int seg = tid & 7;
  const int lrow = tid >> 3;
  int a_iy0[A_PASSES], a_ix0[A_PASSES];
  int64_t a_img[A_PASSES];  // pixel offset of the image start, or -1 if the row is past M
#pragma unroll
  for (int q = 0; q < A_PASSES; ++q) {
    int64_t m = m0 + lrow + q * ROWS_PER_PASS;
    if (m < M) {
      int hw = p.Ho * p.Wo;
      int b = (int)(m / hw);
      int rem = (int)(m - (int64_t)b * hw);
      int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[q] = oy * p.stride - p.pad;
      a_ix0[q] = ox * p.stride - p.pad;
      a_img[q] = (int64_t)b * p.Hi * p.Wi;
    } else {
      a_iy0[q] = 0;
      a_ix0[q] = 0;
      a_img[q] = -1;
    }
  }
  bool b_ok[B_PASSES];
#pragma unroll
  for (int q = 0; q < B_PASSES; ++q) b_ok[q] = (n0 + lrow + q * ROWS_PER_PASS) < p.Cout;

  const int chunks_per_tap = (p.Cin + BK - 1) / BK;
  const int ntaps = p.KH * p.KW;
  const int nchunks = ntaps * chunks_per_tap;

  f32x4 ra[A_PASSES], rb[B_PASSES];

  auto load_chunk = [&](int chunk) {
    int tap = chunk / chunks_per_tap;
    int c0 = (chunk - tap * chunks_per_tap) * BK + seg * 4;
    int kh = tap / p.KW, kw = tap - kh * p.KW;
    bool c_ok = c0 < p.Cin;
    // row-run mode (rr_cin > 0): one "tap" is a whole kernel ROW; its K range is the contiguous run of
    // KW*rr_cin floats that starts at input pixel (iy, ox*stride - pad) - NHWC keeps the pixels of a
    // row adjacent - so small-Cin wide-window convs (the k = 7 / 15 init convs over 12 channels) fill
    // their 32-wide K chunks instead of wasting 20 of 32 lanes per tap.  Validity is per 16-B segment.
    const int kwp = p.rr_cin > 0 ? c0 / p.rr_cin : kw;
#pragma unroll
    for (int q = 0; q < A_PASSES; ++q) {
      int iy = a_iy0[q] + kh, ix = a_ix0[q] + kwp;
      bool ok = c_ok && a_img[q] >= 0 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int ixb = p.rr_cin > 0 ? a_ix0[q] : ix;
      if (ok) v = *(const f32x4*)(p.x + (a_img[q] + (int64_t)iy * p.Wi + ixb) * p.ldx + c0);
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c_ok && b_ok[q])
        v = *(const f32x4*)(p.w + ((int64_t)tap * p.Cout + n0 + lrow + q * ROWS_PER_PASS) * p.Cin + c0);
      rb[q] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int q = 0; q < A_PASSES; ++q)
      *(f32x4*)(As + (lrow + q * ROWS_PER_PASS) * LDS_LD + seg * 4) = ra[q];
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q)
      *(f32x4*)(Bs + (lrow + q * ROWS_PER_PASS) * LDS_LD + seg * 4) = rb[q];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;
  const float* a_base = As + (wm * TM * 32 + frow) * LDS_LD + fk;
  const float* b_base = Bs + (wn * TN * 32 + frow) * LDS_LD + fk;

  load_chunk(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    store_chunk();
    __syncthreads();
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);  // in flight while the MFMAs below run
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(a_base + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(b_base + j * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if (p.out_mode == OUT_NHWC && p.wide_epilogue && (!p.gate_src || ((p.Ho * p.Wo) & 31) == 0)) {
    static_assert(LDS_FLOATS >= WAVES_M * WAVES_N * 1024, "the 16-byte epilogue needs 4 KB of LDS per wave");
    if (p.seg_partial) store_tile_regs_wide<BM, BN, WAVES_M, WAVES_N, true>(p, lds, acc, m0, n0, M);
    else store_tile_regs_wide<BM, BN, WAVES_M, WAVES_N, false>(p, lds, acc, m0, n0, M);
  } else if (p.out_mode == OUT_NHWC && (!p.gate_src || (((p.Ho * p.Wo) & 31) == 0 && !p.res && p.act == ACT_NONE)))
    store_tile_regs<BM, BN, WAVES_M, WAVES_N>(p, acc, m0, n0, M);
  else
    store_tile<BM, BN, WAVES_M, WAVES_N>(p, lds, acc, m0, n0, M);
}

// ------------------------------------------------------------------------------------------------
// Fast path (Cin % 32 == 0): operands go global -> LDS directly with `buffer_load_dwordx4 ... lds`
// (no VGPR staging, no ds_write).  Per lane only a 32-bit byte offset is kept (fixed per tap for the
// activations, fixed for the whole kernel for the weights); the advance along K is a SCALAR offset
// and padding pixels / rows past the edge use an out-of-range offset (the buffer range check returns
// 0), so a K-chunk costs no vector ALU for addressing.  LDS rows are 128 B unpadded (one
// wave-instruction writes 1 KiB linearly = 8 rows); ds_read_b128 bank conflicts are removed by an XOR
// swizzle applied to the SOURCE segment (lane holding LDS slot s of row r fetches 16-B segment
// s ^ ((r>>1)&7)) and to the READ address.  Two LDS stages, static stage index (loop unrolled by 2),
// counted vmcnt + raw s_barrier so the next chunk's DMA stays in flight across the barrier.
// Measured on MI355X (profiles/): 113-133 TFLOP/s vs 97-116 for the register-staged kernel.
constexpr uint32_t OOB_OFF = 0x80000000u;

template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_buf_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist on the device side only
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int ROWS_PER_PASS = NT / 8;
  constexpr int A_PASSES = BM / ROWS_PER_PASS;
  constexpr int B_PASSES = BN / ROWS_PER_PASS;
  constexpr int NLOADS = A_PASSES + B_PASSES;
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int STAGE = (BM + BN) * BK;  // floats
  constexpr int C_LD = WAVES_N * 32 + 4;
  constexpr int LDS_FLOATS = 2 * STAGE > BM * C_LD ? 2 * STAGE : BM * C_LD;
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  // Tile order: the N tiles of one M tile run back to back on ONE XCD (workgroups are dealt round-robin to
  // the 8 XCDs in dispatch order, x fastest), so the activation rows they share are read from HBM once and
  // hit that XCD's L2 afterwards.  In plain (x, y) order all M tiles of N tile 0 run before N tile 1 and a
  // map larger than the caches is streamed from HBM once per N tile.
  int mt = blockIdx.x, nt = blockIdx.y;
  if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
    const unsigned lid = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned slot = lid >> 3;
    nt = slot % gridDim.y;
    mt = (slot / gridDim.y) * 8 + (lid & 7);
  }
  const int64_t m0 = (int64_t)mt * BM;
  const int n0 = nt * BN;
  const int lrow = tid >> 3;
  const int gseg = (tid & 7) ^ ((lrow >> 1) & 7);

  // activations: buffer based at the image of the tile's first pixel (keeps offsets < 2^31); a pure
  // GEMM (1x1, stride 1, no padding) has contiguous rows and is based at the tile's first row
  // instead, so tensors of any size work (the Winograd V/D matrices exceed 2 GiB)
  const bool gemm = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
  const int hw = p.Ho * p.Wo;
  const int img0 = gemm ? 0 : (int)(m0 / hw);
  const int64_t img_elems = (int64_t)p.Hi * p.Wi * p.ldx;
  const int64_t a_total = gemm ? (M - m0 < BM ? M - m0 : (int64_t)BM) * p.ldx * 4 : ((int64_t)p.B - img0) * img_elems * 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (gemm ? m0 * p.ldx : (int64_t)img0 * img_elems)), 0,
      (int)(a_total > 0x7fffffff ? 0x7fffffff : a_total), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w, 0, (int)((int64_t)(p.wz_rows > 0 ? p.wz_count : p.KH * p.KW) * p.Cout * p.Cin * 4), 0x00020000);
  // batched GEMM: the whole tile lies in one weight slab (wz_rows % BM == 0)
  const uint32_t wz_off = p.wz_rows > 0 ? (uint32_t)(m0 / p.wz_rows) * (uint32_t)(p.Cout * p.Cin * 4) : 0u;

  int a_iy0[A_PASSES], a_ix0[A_PASSES];
  int a_img[A_PASSES];  // pixel offset of the row's image relative to img0, or -1 past the edge
#pragma unroll
  for (int q = 0; q < A_PASSES; ++q) {
    int64_t m = m0 + lrow + q * ROWS_PER_PASS;
    if (m < M && gemm) {
      a_iy0[q] = 0;
      a_ix0[q] = 0;
      a_img[q] = lrow + q * ROWS_PER_PASS;
    } else if (m < M) {
      int b = (int)(m / hw);
      int rem = (int)(m - (int64_t)b * hw);
      int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[q] = oy * p.stride - p.pad;
      a_ix0[q] = ox * p.stride - p.pad;
      a_img[q] = (b - img0) * p.Hi * p.Wi;
    } else {
      a_iy0[q] = 0;
      a_ix0[q] = 0;
      a_img[q] = -1;
    }
  }
  uint32_t voffB[B_PASSES];
#pragma unroll
  for (int q = 0; q < B_PASSES; ++q) {
    int n = n0 + lrow + q * ROWS_PER_PASS;
    voffB[q] = n < p.Cout ? (uint32_t)((n * p.Cin + gseg * 4) * 4) : OOB_OFF;
  }
  uint32_t voffA[A_PASSES];
  const int chunks_per_tap = p.Cin / BK;
  const int nchunks = p.KH * p.KW * chunks_per_tap;
  // split-K (small-M layers that cannot fill the chip with tiles): workgroup z sums the chunks
  // [c_begin, c_end) and writes a raw partial tile; conv_splitk_reduce_kernel adds the partials in a
  // fixed order and applies the epilogue
  int c_begin = 0, c_end = nchunks;
  if (p.ksplit > 1) {
    const int per = (nchunks + p.ksplit - 1) / p.ksplit;
    c_begin = (int)blockIdx.z * per;
    c_end = c_begin + per < nchunks ? c_begin + per : nchunks;
  }
  int cur_tap = c_begin / chunks_per_tap - 1, cur_cc = chunks_per_tap;
  uint32_t soffA = 0, soffB = 0;
  const uint32_t tap_stride_b = (uint32_t)p.Cout * p.Cin * 4;

  auto next_tap = [&]() {
    ++cur_tap;
    cur_cc = 0;
    int kh = cur_tap / p.KW, kw = cur_tap - kh * p.KW;
#pragma unroll
    for (int q = 0; q < A_PASSES; ++q) {
      int iy = a_iy0[q] + kh, ix = a_ix0[q] + kw;
      bool ok = a_img[q] >= 0 && (gemm || (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi));
      int pix = gemm ? a_img[q] : a_img[q] + iy * p.Wi + ix;
      voffA[q] = ok ? (uint32_t)((pix * p.ldx + gseg * 4) * 4) : OOB_OFF;
    }
    soffA = 0;
    soffB = (uint32_t)cur_tap * tap_stride_b + wz_off;
  };
  auto issue = [&](float* stage_base) {  // DMA of the next K-chunk into the given stage
    if (cur_cc == chunks_per_tap) next_tap();
    __attribute__((address_space(3))) float* sb =
        (__attribute__((address_space(3))) float*)(stage_base + wave * 8 * BK);
    // the scalar offsets are wave-uniform, but hipcc keeps them in VGPRs and would wrap every load in a
    // readfirstlane waterfall loop (10 instructions + exec juggling per load): pin them to SGPRs
    const uint32_t sa = __builtin_amdgcn_readfirstlane(soffA), sbo = __builtin_amdgcn_readfirstlane(soffB);
#pragma unroll
    for (int q = 0; q < A_PASSES; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, sb + q * ROWS_PER_PASS * BK, 16, voffA[q], sa, 0, 0);
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, sb + BM * BK + q * ROWS_PER_PASS * BK, 16, voffB[q], sbo, 0, 0);
    soffA += BK * 4;
    soffB += BK * 4;
    ++cur_cc;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  const int khalf = lane >> 5;
  const int a_row = (wm * TM * 32 + frow) * BK;
  const int b_row = BM * BK + (wn * TN * 32 + frow) * BK;
  // (reading the fragments of k-group kk+1 ahead of the MFMAs of group kk - two register sets and a
  //  sched_barrier so that hipcc does not sink the ds_reads - was measured: no change, the partner wave
  //  of the SIMD already covers the LDS latency)
  auto compute = [&](const float* base) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int slot = ((2 * kk + khalf) ^ fsw) * 4;
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(base + a_row + i * 32 * BK + slot);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(base + b_row + j * 32 * BK + slot);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
  };

  float* const s0 = lds;
  float* const s1 = lds + STAGE;
  if (c_begin > 0) {  // start in the middle of a tap
    next_tap();
    cur_cc = c_begin - cur_tap * chunks_per_tap;
    soffA = (uint32_t)cur_cc * BK * 4;
    soffB += (uint32_t)cur_cc * BK * 4;
  }
  issue(s0);
  // Two chunks per trip with static stage addresses and ONE loop exit: with a `break` between the two
  // halves hipcc kept two copies of the 64 accumulator registers and moved one onto the other on every
  // trip (32 v_mov_b64 behind a drained MFMA pipe).  An odd last chunk is handled after the loop.
  int c = c_begin;
  for (; c + 1 < c_end; c += 2) {
    issue(s1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    compute(s0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < c_end) {
      issue(s0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    compute(s1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (c < c_end) {  // odd chunk count: the last chunk is in stage 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(s0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __syncthreads();
  if (p.ksplit > 1) {  // raw partial sums [z][M][Cout]
    ConvParams q = p;
    q.bias = nullptr; q.act = ACT_NONE; q.res = nullptr; q.gate_src = nullptr; q.gate = nullptr;
    q.out_mode = OUT_NHWC; q.ldy = p.Cout; q.yoff = 0;
    q.y = p.partial + (int64_t)blockIdx.z * M * p.Cout;
    q.seg_partial = nullptr;
    if (p.wide_epilogue) store_tile_regs_wide<BM, BN, WAVES_M, WAVES_N, false>(q, lds, acc, m0, n0, M);
    else store_tile_regs<BM, BN, WAVES_M, WAVES_N>(q, acc, m0, n0, M);
  } else if (p.out_mode == OUT_NHWC && p.wide_epilogue && (!p.gate_src || (hw & 31) == 0)) {
    if (p.seg_partial) store_tile_regs_wide<BM, BN, WAVES_M, WAVES_N, true>(p, lds, acc, m0, n0, M);
    else store_tile_regs_wide<BM, BN, WAVES_M, WAVES_N, false>(p, lds, acc, m0, n0, M);
  } else if (p.out_mode == OUT_NHWC && (!p.gate_src || ((hw & 31) == 0 && !p.res && p.act == ACT_NONE))) {
    store_tile_regs<BM, BN, WAVES_M, WAVES_N>(p, acc, m0, n0, M);
  } else if (p.out_mode == OUT_PIXSHUF && ((p.Cout >> 2) & 31) == 0 && (p.Wo & 31) == 0 && !p.gate_src && !p.res &&
             p.wide_epilogue) {
    if (p.seg_partial) store_tile_regs_pixshuf_wide<BM, BN, WAVES_M, WAVES_N, true>(p, lds, acc, m0, n0, M);
    else store_tile_regs_pixshuf_wide<BM, BN, WAVES_M, WAVES_N, false>(p, lds, acc, m0, n0, M);
  } else if (p.out_mode == OUT_PIXSHUF && ((p.Cout >> 2) & 31) == 0 && (p.Wo & 31) == 0 && !p.gate_src && !p.res) {
    if (p.seg_partial)
      store_tile_regs_pixshuf<BM, BN, WAVES_M, WAVES_N, true>(p, acc, m0, n0, M);
    else
      store_tile_regs_pixshuf<BM, BN, WAVES_M, WAVES_N, false>(p, acc, m0, n0, M);
  } else {
    store_tile<BM, BN, WAVES_M, WAVES_N>(p, lds, acc, m0, n0, M);
  }
#endif
}

// y[m][n..n+3] = epilogue( sum_z partial[z][m][n..n+3] ), z ascending (deterministic)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvParams p, int64_t M) {
  const int C4 = p.Cout >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C4) return;
  const int64_t m = idx / C4;
  const int n = (int)(idx - m * C4) * 4;
  const int64_t zstride = M * p.Cout;
  const float* src = p.partial + m * p.Cout + n;
  f32x4 s = *(const f32x4*)src;
  int z = 1;
  for (; z + 3 < p.ksplit; z += 4) {   // four partials in flight, added in z order (the same sum as one at a time)
    const f32x4 v0 = *(const f32x4*)(src + z * zstride), v1 = *(const f32x4*)(src + (z + 1) * zstride);
    const f32x4 v2 = *(const f32x4*)(src + (z + 2) * zstride), v3 = *(const f32x4*)(src + (z + 3) * zstride);
    s += v0;
    s += v1;
    s += v2;
    s += v3;
  }
  for (; z < p.ksplit; ++z) s += *(const f32x4*)(src + z * zstride);
  const bool vec_ok = p.out_mode != OUT_NCHW && (p.ldy & 3) == 0 && (p.yoff & 3) == 0 && (((uintptr_t)p.y) & 15) == 0 &&
                      (!p.res || ((p.ldres & 3) == 0 && (((uintptr_t)p.res) & 15) == 0)) &&
                      (!p.gate_src || (p.ldgs & 3) == 0) && (p.out_mode != OUT_PIXSHUF || (p.Cout & 15) == 0);
  if (vec_ok) {
    float v[4] = {s[0], s[1], s[2], s[3]};
    epilogue_store<4>(p, m, n, v);
    if (p.seg_partial) {
      // GroupNorm partials of the finished rows (conv_seg_chunks: NHWC, Cout % 16 == 0): one chunk per output pixel, the
      // four lanes that hold a 16-channel segment of a row are adjacent (idx = m C4 + n / 4, C4 % 4 == 0)
      double d1 = (double)((v[0] + v[1]) + (v[2] + v[3]));
      double d2 = (double)fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
      d1 += __shfl_xor(d1, 1, 64);
      d2 += __shfl_xor(d2, 1, 64);
      d1 += __shfl_xor(d1, 2, 64);
      d2 += __shfl_xor(d2, 2, 64);
      if ((n & 15) == 0) {
        const int hw_o = p.Ho * p.Wo;
        const int b = (int)(m / hw_o);
        const int seg = (p.yoff + n - p.seg_c0) >> 4;
        double* op = p.seg_partial + (((int64_t)b * p.seg_nseg + seg) * hw_o + (m - (int64_t)b * hw_o)) * 2;
        op[0] = d1;
        op[1] = d2;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[1] = {s[e]};
      epilogue_store<1>(p, m, n + e, v);
    }
  }
}

int64_t conv_macs(const ConvParams& p) {
  return (int64_t)p.B * p.Ho * p.Wo * p.Cout * p.Cin * p.KH * p.KW;
}

// Shape conditions of the buffer-DMA fast path (pointer alignment is checked at launch).
static bool fast_shape_ok(const ConvParams& p) {
  const int64_t w_bytes = (int64_t)(p.wz_rows > 0 ? p.wz_count : p.KH * p.KW) * p.Cout * p.Cin * 4;
  // the fast kernel addresses the activations with 32-bit byte offsets relative to the image of the
  // tile's first pixel: every image a 256-row tile can touch must lie within 2^31 bytes of it
  const int64_t img_bytes = (int64_t)p.Hi * p.Wi * p.ldx * 4;
  const int64_t hw_o = (int64_t)p.Ho * p.Wo;
  const int64_t span = (hw_o % 256 == 0) ? 1 : 255 / hw_o + 2;
  const bool gemm = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;  // based per tile: no size limit
  return p.rr_cin == 0 && (p.Cin % BK) == 0 && p.Cout > 32 && w_bytes < 0x7fffffff &&
         (gemm || span * img_bytes < 0x7fffffff);
}

// Number of K splits for a conv whose M x Cout tiles cannot fill the chip (batch-1 patches of the
// ultra-res grid: M = 64 .. 1024 pixels against K = 9 x 2048): enough workgroups for ~3 per CU, at
// least 4 chunks each.  1 = no split.  The caller provides ConvParams::partial [ksplit][M][Cout].
int conv_ksplit(const ConvParams& p) {
  if (!fast_shape_ok(p) || p.wz_rows > 0 || p.Cout < 64 || (p.Cout & 3)) return 1;
  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  // (M <= 64 runs 64 x 128 tiles, see launch_conv_igemm: half as many tiles as 128 x 64 counts here)
  const int64_t tiles = M <= 64 ? (p.Cout + 127) / 128 : ((M + 127) / 128) * ((p.Cout + 63) / 64);
  const int nchunks = p.KH * p.KW * (p.Cin / BK);
  if (tiles >= 256 || nchunks < 16) return 1;
  const int want = (int)((768 + tiles - 1) / tiles);
  int per = (nchunks + want - 1) / want;
  if (per < 4) per = 4;
  return (nchunks + per - 1) / per;
}

// Chunks per image of the GroupNorm partials this launch can leave in its epilogue (ConvParams::seg_partial), or 0
// when its epilogue cannot (split-K, the LDS-staged epilogue, ragged shapes): the plan then keeps the statistics pass.
int conv_seg_chunks(const ConvParams& p) {
  const int64_t hw = (int64_t)p.Ho * p.Wo, M = (int64_t)p.B * hw;
  if (p.partial && conv_ksplit(p) > 1)   // split-K: the reduction kernel leaves one chunk per output pixel (16-byte path)
    return (p.out_mode == OUT_NHWC && !(p.Cout & 15) && !((p.yoff - p.seg_c0) & 15) && !(p.ldy & 3) && !(p.yoff & 3) &&
            (!p.res || (!(p.ldres & 3) && !((uintptr_t)p.res & 15))) && (!p.gate_src || !(p.ldgs & 3)) && hw < 0x7fffffff)
               ? (int)hw
               : 0;
  if ((hw & 31) || (p.Cout & 15) || ((p.yoff - p.seg_c0) & 15)) return 0;
  if (p.out_mode == OUT_NHWC) {
    if (p.act != ACT_NONE || (p.gate_src && p.res)) return 0;
    return (int)(hw >> 5);
  }
  if (p.out_mode == OUT_PIXSHUF) {
    const bool fast = fast_shape_ok(p) && M > 64;
    if (!fast || ((p.Cout >> 2) & 31) || (p.Wo & 31) || p.gate_src || p.res) return 0;
    return (int)(hw >> 5) * 4;
  }
  return 0;
}

// Tile shape of the buffer-DMA kernel for a fast-path shape (fast_shape_ok)
static int launch_conv_fast(const ConvParams& p, int64_t M, hipStream_t s) {
  const int64_t tiles256 = ((M + 255) / 256) * ((p.Cout + 127) / 128);
  // 256x128 (one 8-wave workgroup per CU) needs a K loop long enough to amortise its serial
  // prologue/epilogue; short-K 1x1 convs run as two 128x128 workgroups per CU, which overlap one
  // tile's epilogue with the other's loop (measured in the step: 3x3 Cin=128 layers still prefer 256x128)
  const int k_chunks = p.KH * p.KW * (p.Cin / BK);
  static const int shortk = kd_switch("KD_SHORTK_TILE", 0);   // experiment switch
  if (p.wz_rows > 0) {
    // Batched position GEMMs of the Winograd layers: M = positions x tiles is a few thousand rows, so the launch is a
    // handful of tile rounds and the last, partly filled round decides its time (36 x 256 rows x 1024 columns: 576
    // tiles of 128 x 128 on 512 slots = two rounds for 1.1 rounds of work).  Pick the tile shape whose rounds are
    // fullest, larger tiles winning ties by their better operand reuse (weights: per-tile efficiency seen in the step)
    static int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      int dev = 0;
      KD_HIP_CHECK(hipGetDevice(&dev));
      KD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    struct Cand { int bm, bn, per_cu; double w; };
    const Cand cands[3] = {{256, 128, 1, 1.00}, {128, 128, 2, 0.97}, {128, 64, 3, 0.92}};
    int best = 2;
    double best_eff = 0.0;
    for (int i = 0; i < 3; ++i) {
      if (p.wz_rows % cands[i].bm) continue;   // a tile must lie in one weight slab
      const int64_t tiles = ((M + cands[i].bm - 1) / cands[i].bm) * ((p.Cout + cands[i].bn - 1) / cands[i].bn);
      const int64_t slots = (int64_t)cus * cands[i].per_cu;
      const double eff = cands[i].w * (double)tiles / (double)(((tiles + slots - 1) / slots) * slots);
      if (eff > best_eff + 1e-9) {
        best_eff = eff;
        best = i;
      }
    }
    if (kd_switch("KD_WZ_TILE", 0) > 0 && p.wz_rows % 256 == 0) best = kd_switch("KD_WZ_TILE", 0) - 1;   // experiment: force a shape
    if (best == 0) {
      dim3 grid((unsigned)((M + 255) / 256), (p.Cout + 127) / 128);
      hipLaunchKernelGGL((conv_buf_kernel<256, 128, 4, 2, 2>), grid, dim3(512), 0, s, p);
    } else if (best == 1) {
      dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 127) / 128);
      hipLaunchKernelGGL((conv_buf_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, s, p);
    } else {
      dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 63) / 64);
      hipLaunchKernelGGL((conv_buf_kernel<128, 64, 2, 2, 3>), grid, dim3(256), 0, s, p);
    }
  } else if (p.KH * p.KW == 1 && k_chunks <= 8 && p.wz_rows == 0 && shortk == 1 &&
      ((M + 127) / 128) * ((p.Cout + 127) / 128) >= 256) {
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_buf_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, s, p);
  } else if (p.KH * p.KW == 1 && k_chunks <= 8 && p.wz_rows == 0 && shortk == 2 && tiles256 >= 256) {
    dim3 grid((unsigned)((M + 255) / 256), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_buf_kernel<256, 128, 4, 2, 2>), grid, dim3(512), 0, s, p);
  } else if (p.KH * p.KW == 1 && k_chunks <= 8 && p.wz_rows == 0) {
    // 1x1 convs with K <= 256 (ResnetBlock skip convs with the gate epilogue, pixel-shuffle upsamples)
    // move as many bytes as they compute: three 128x64 workgroups per CU keep more loads in flight
    // (measured -12..16 % on them; the Winograd GEMMs of the same K lose 4 % with this shape)
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 63) / 64);
    hipLaunchKernelGGL((conv_buf_kernel<128, 64, 2, 2, 3>), grid, dim3(256), 0, s, p);
  } else if (tiles256 >= 512 && (p.KH * p.KW > 1 || k_chunks >= 64)) {
    dim3 grid((unsigned)((M + 255) / 256), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_buf_kernel<256, 128, 4, 2, 2>), grid, dim3(512), 0, s, p);
  } else if (((M + 127) / 128) * ((p.Cout + 127) / 128) >= 512) {
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_buf_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, s, p);
  } else {  // few tiles (deep 16x16 levels, attention projections): 128x64 tiles, 3 workgroups per CU
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 63) / 64);
    hipLaunchKernelGGL((conv_buf_kernel<128, 64, 2, 2, 3>), grid, dim3(256), 0, s, p);
  }
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// 16-byte epilogue (store_tile_regs_wide): NHWC output whose rows, and those of every map the epilogue adds, start on
// 16-byte boundaries; whole 32-row MFMA tiles (and one image per tile where a per-image gate is applied)
static bool conv_wide_ok(const ConvParams& p) {
  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  return p.ksplit <= 1 && M % 32 == 0 && p.Cout % 4 == 0 && p.ldy % 4 == 0 && p.yoff % 4 == 0 && al(p.y) && al(p.bias) &&
         (!p.res || (al(p.res) && p.ldres % 4 == 0)) &&
         (!p.gate_src || (al(p.gate_src) && al(p.gate) && p.ldgs % 4 == 0 && (p.Ho * p.Wo) % 32 == 0)) &&
         (!p.seg_partial || ((p.Ho * p.Wo) % 32 == 0 && p.Cout % 16 == 0));
}

int launch_conv_igemm(const ConvParams& p, hipStream_t s) {
  KD_REQUIRE(p.Cin % 4 == 0 && p.ldx % 4 == 0, "igemm needs Cin and ldx multiples of 4");
  KD_REQUIRE(!p.seg_partial || (conv_seg_chunks(p) > 0 && p.seg_nseg > 0),
             "conv: this launch cannot leave GroupNorm partials (see conv_seg_chunks)");
  KD_REQUIRE(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "igemm operands must be 16-B aligned");
  if (p.out_mode == OUT_PIXSHUF) KD_REQUIRE(p.Cout % 4 == 0, "pixel-shuffle needs Cout % 4 == 0");
  int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  KD_REQUIRE(M > 0 && p.Cout > 0, "empty conv");
  if (p.wz_rows > 0)
    KD_REQUIRE(p.KH * p.KW == 1 && p.wz_rows % 128 == 0 && M == (int64_t)p.wz_rows * p.wz_count,
               "batched GEMM: 1x1 only, slab rows a multiple of 128, M = rows x slabs");
  const int ks = p.partial ? conv_ksplit(p) : 1;
  if (ks > 1) {
    KD_REQUIRE(((uintptr_t)p.partial & 15) == 0, "split-K partial buffer must be 16-B aligned");
    KD_REQUIRE(!p.seg_partial || ((((uintptr_t)p.y | (uintptr_t)p.res) & 15) == 0),
               "split-K conv: GroupNorm partials come from the 16-byte path of the reduction (aligned output / residual)");
    ConvParams q = p;
    q.ksplit = ks;
    q.wide_epilogue = p.Cout % 4 == 0 &&   // the raw partial tiles [z][M][Cout] with 16-byte stores
                      kd_switch("KD_CONV_WIDE", 1) != 0;
    if (M <= 64) {
      // an 8x8 map of a batch-1 patch: 64-row tiles (a 128-row tile would spend half its MFMAs on padding and
      // these launches, K up to 9 x 3072, were bound by exactly that: 124 us for 226 MB of weights), 128 columns
      dim3 grid(1, (p.Cout + 127) / 128, ks);
      hipLaunchKernelGGL((conv_buf_kernel<64, 128, 1, 4, 3>), grid, dim3(256), 0, s, q);
    } else {
      dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 63) / 64, ks);
      hipLaunchKernelGGL((conv_buf_kernel<128, 64, 2, 2, 3>), grid, dim3(256), 0, s, q);
    }
    const int64_t total = M * (p.Cout / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, M);
    KD_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const bool fast = fast_shape_ok(p) && M > 64;
  if (fast && (p.out_mode == OUT_NHWC || p.out_mode == OUT_PIXSHUF) && conv_wide_ok(p) &&
      kd_switch("KD_CONV_WIDE", 1) != 0) {   // (A/B switch)
    ConvParams q = p;
    q.wide_epilogue = 1;
    return launch_conv_fast(q, M, s);
  }
  if (fast) return launch_conv_fast(p, M, s);
  KD_REQUIRE(p.wz_rows == 0, "batched GEMM needs the buffer-load fast path (Cin % 32 == 0, Cout > 32)");
  ConvParams g = p;
  g.wide_epilogue = p.out_mode == OUT_NHWC && conv_wide_ok(p) && kd_switch("KD_CONV_WIDE", 1) != 0;
  if (p.Cout <= 32) {
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 31) / 32);
    hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1>), grid, dim3(256), 0, s, g);
  } else if (p.Cout <= 64) {
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 63) / 64);
    hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2>), grid, dim3(256), 0, s, g);
  } else if (M <= 64) {
    dim3 grid((unsigned)((M + 63) / 64), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_igemm_kernel<64, 128, 1, 4>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), (p.Cout + 127) / 128);
    hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, g);
  }
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------ weight packing
__global__ void pack_oihw_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int Ireal, int I,
                                 int KH, int KW) {
  int64_t total = (int64_t)O * I * KH * KW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    // out index: ((tap*O + o)*I + i)
    int i = (int)(idx % I);
    int64_t t = idx / I;
    int o = (int)(t % O);
    int tap = (int)(t / O);
    out[idx] = i < Ireal ? w[((int64_t)o * Ireal + i) * KH * KW + tap] : 0.f;
  }
}
int launch_pack_oihw(const float* w, float* out, int O, int Ireal, int I, int KH, int KW, hipStream_t s) {
  int64_t total = (int64_t)O * I * KH * KW;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_oihw_kernel, dim3(blocks), dim3(256), 0, s, w, out, O, Ireal, I, KH, KW);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void pack_oihw_rowrun_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int Ireal, int I,
                                        int KH, int KW) {
  int64_t total = (int64_t)O * I * KH * KW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    // out index: ((kh*O + o)*KW + kw)*I + i
    int i = (int)(idx % I);
    int64_t t = idx / I;
    int kw = (int)(t % KW);
    t /= KW;
    int o = (int)(t % O);
    int kh = (int)(t / O);
    out[idx] = i < Ireal ? w[(((int64_t)o * Ireal + i) * KH + kh) * KW + kw] : 0.f;
  }
}
int launch_pack_oihw_rowrun(const float* w, float* out, int O, int Ireal, int I, int KH, int KW, hipStream_t s) {
  int64_t total = (int64_t)O * I * KH * KW;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_oihw_rowrun_kernel, dim3(blocks), dim3(256), 0, s, w, out, O, Ireal, I, KH, KW);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// Row-run packing of a SUBSET of the input channels: packed channel i comes from source channel
// a0+i (i < na) or b0+(i-na) (i < na+nb), zero above; Itot = channels of the OIHW source.
__global__ void pack_oihw_rowrun_sub_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int Itot,
                                            int a0, int na, int b0, int nb, int I, int KH, int KW) {
  int64_t total = (int64_t)O * I * KH * KW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(idx % I);
    int64_t t = idx / I;
    int kw = (int)(t % KW);
    t /= KW;
    int o = (int)(t % O);
    int kh = (int)(t / O);
    int src = i < na ? a0 + i : (i < na + nb ? b0 + (i - na) : -1);
    out[idx] = src >= 0 ? w[(((int64_t)o * Itot + src) * KH + kh) * KW + kw] : 0.f;
  }
}
int launch_pack_oihw_rowrun_sub(const float* w, float* out, int O, int Itot, int a0, int na, int b0, int nb, int I,
                                int KH, int KW, hipStream_t s) {
  int64_t total = (int64_t)O * I * KH * KW;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_oihw_rowrun_sub_kernel, dim3(blocks), dim3(256), 0, s, w, out, O, Itot, a0, na, b0, nb, I,
                     KH, KW);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void pack_unshuffle_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int C) {
  int64_t total = (int64_t)O * C * 4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % C);
    int64_t t = idx / C;
    int o = (int)(t % O);
    int tap = (int)(t / O);  // s1*2+s2
    out[idx] = w[(int64_t)o * 4 * C + c * 4 + tap];
  }
}
int launch_pack_unshuffle(const float* w, float* out, int O, int C, hipStream_t s) {
  int64_t total = (int64_t)O * C * 4;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_unshuffle_kernel, dim3(blocks), dim3(256), 0, s, w, out, O, C);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void pack_shuffle_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ wout, float* __restrict__ bout, int Co, int I) {
  int64_t total = (int64_t)4 * Co * I;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(idx % I);
    int np = (int)(idx / I);  // packed row (q*Co + c)
    int q = np / Co, c = np - q * Co;
    wout[idx] = w[((int64_t)c * 4 + q) * I + i];
    if (i == 0) bout[np] = b[c * 4 + q];
  }
}
int launch_pack_shuffle(const float* w, const float* b, float* wout, float* bout, int Co, int I, hipStream_t s) {
  int64_t total = (int64_t)4 * Co * I;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_shuffle_kernel, dim3(blocks), dim3(256), 0, s, w, b, wout, bout, Co, I);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
