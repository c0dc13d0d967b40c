// Winograd F(2x2,3x3) transforms for the deep 3x3 convolutions of the ResnetBlocks.
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; the same identity cuDNN /
//                                                          MIOpen select for fp32 3x3 stride-1 convs)
//
// A 3x3 / stride 1 / pad 1 conv over [B,H,W,Cin] becomes 16 independent GEMMs
// D_p[t][n] = sum_c V_p[t][c] * U_p[n][c], one per position p of the 4x4 transformed tile, over the
// Mt = B*(H/2)*(W/2) output tiles: 16*Mt*Cin*Cout MACs instead of 36*Mt*Cin*Cout (2.25x fewer MFMA
// issues).  The GEMMs run on the fast implicit-GEMM kernel (kernels_conv.hip, weight slab selected
// by tile row: ConvParams::wz_rows); this file holds the three HBM-bound transform kernels:
//
//   wino_pack  (plan build)  OIHW weights            -> U [16][Cout][Cin]
//   wino_in    (per step)    GroupNorm+FiLM+SiLU(x)  -> V [16][Mt][Cin]   (replaces gn_apply: the
//                                                       normalised map is never materialised)
//   wino_out   (per step)    D [16][Mt][Cout]        -> y NHWC, + bias (+ residual)
//
// The transforms move 4x the feature map, so the plan uses this path only where the GEMM saving
// outweighs that traffic (measured: Cin >= 256 wins, Cin = 128 loses); everything is fp32, the
// result differs from the direct conv by re-association only (tests: <= 4e-6 relative L2).
#include "common.h"

namespace kd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wino_silu(float v) { return v / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int O,
                                                        int I) {
  const int64_t total = (int64_t)O * I;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const float* g = w + idx * 9;
    float t[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float g0 = g[k], g1 = g[3 + k], g2 = g[6 + k];
      t[0][k] = g0;
      t[1][k] = 0.5f * (g0 + g1 + g2);
      t[2][k] = 0.5f * (g0 - g1 + g2);
      t[3][k] = g2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
      for (int s = 0; s < 4; ++s) U[(int64_t)(r * 4 + s) * total + idx] = u[s];
    }
  }
}

int launch_wino_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s) {
  int64_t total = (int64_t)O * I;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, U, O, I);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// One thread: one output tile x 4 channels.  Threads run along channels first (16-B coalesced).
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ stats,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ scale_shift, int ld_ss,
                                                      float* __restrict__ V, int B, int H, int W, int C, int G,
                                                      int64_t t0, int64_t nt) {
  const int C4 = C >> 2;
  const int Ht = H >> 1, Wt = W >> 1;
  const int64_t Mt = nt;  // tiles of this slice: V is [16][nt][C]
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mt * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t tl = idx / C4;
  const int64_t t = t0 + tl;
  const int tx = (int)(t % Wt);
  const int ty = (int)((t / Wt) % Ht);
  const int b = (int)(t / ((int64_t)Wt * Ht));

  float A[4], Bc[4];
  const bool norm = stats != nullptr;
  if (norm) {
    const int Cg = C / G;
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // same folding as gn_apply_silu_kernel (kernels_norm.hip)
      int c = c4 * 4 + e;
      int g = c / Cg;
      float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
      float a = rstd * gamma[c];
      float bb = beta[c] - mean * a;
      if (scale_shift) {
        float sc = scale_shift[(int64_t)b * ld_ss + c] + 1.0f;
        float sh = scale_shift[(int64_t)b * ld_ss + C + c];
        a *= sc;
        bb = bb * sc + sh;
      }
      A[e] = a;
      Bc[e] = bb;
    }
  }

  // the 16 patch values first, unconditionally (coordinates clamped into the image; what lies outside is zeroed after the
  // activation - the conv pads the ACTIVATED map): with every load in its own bounds branch hipcc waited for each before the
  // next (round 5, as in wino4_in_kernel)
  f32x4 d[4][4];
  bool rok[4], cok[4];
  int64_t rowp[4];
  int colp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int iy = 2 * ty - 1 + r;
    rok[r] = iy >= 0 && iy < H;
    rowp[r] = ((int64_t)b * H + (iy < 0 ? 0 : iy >= H ? H - 1 : iy)) * W;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ix = 2 * tx - 1 + s;
    cok[s] = ix >= 0 && ix < W;
    colp[s] = ix < 0 ? 0 : ix >= W ? W - 1 : ix;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) d[r][s] = *(const f32x4*)(x + (rowp[r] + colp[s]) * ldx + c4 * 4);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 v = d[r][s];
      if (norm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wino_silu(v[e] * A[e] + Bc[e]);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      d[r][s] = (rok[r] && cok[s]) ? v : z;
    }
  // B^T d
  f32x4 u[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u[0][s] = d[0][s] - d[2][s];
    u[1][s] = d[1][s] + d[2][s];
    u[2][s] = d[2][s] - d[1][s];
    u[3][s] = d[1][s] - d[3][s];
  }
  // (B^T d) B
  float* out = V + tl * C + c4 * 4;
  const int64_t pstride = Mt * C;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *(f32x4*)(out + (int64_t)(r * 4 + 0) * pstride) = u[r][0] - u[r][2];
    *(f32x4*)(out + (int64_t)(r * 4 + 1) * pstride) = u[r][1] + u[r][2];
    *(f32x4*)(out + (int64_t)(r * 4 + 2) * pstride) = u[r][2] - u[r][1];
    *(f32x4*)(out + (int64_t)(r * 4 + 3) * pstride) = u[r][1] - u[r][3];
  }
}

int launch_wino_in(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                   const float* scale_shift, int ld_ss, float* V, int B, int H, int W, int C, int G, int64_t t0,
                   int64_t nt, hipStream_t s) {
  KD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && H % 2 == 0 && W % 2 == 0, "Winograd input transform needs even H, W and C % 4 == 0");
  KD_REQUIRE(t0 >= 0 && nt > 0 && t0 + nt <= (int64_t)B * (H / 2) * (W / 2), "tile slice out of range");
  int64_t total = nt * (C / 4);
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, stats, gamma,
                     beta, scale_shift, ld_ss, V, B, H, W, C, G, t0, nt);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// seg != nullptr (C % 16 == 0): fp64 (sum, sum of squares) of the tile's 4 outputs x 16 channels of every 16-channel segment,
// entry [b][c / 16][tile of the image][2] - the GroupNorm partials of y for the layer that normalises it next (SegSrc)
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ D, const float* __restrict__ bias,
                                                       const float* __restrict__ res, int ldres,
                                                       float* __restrict__ y, int B, int H, int W, int C,
                                                       int64_t t0, int64_t nt, double* __restrict__ seg) {
  const int C4 = C >> 2;
  const int Ht = H >> 1, Wt = W >> 1;
  const int64_t Mt = nt;  // tiles of this slice: D is [16][nt][C]
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mt * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t tl = idx / C4;
  const int64_t t = t0 + tl;
  const int tx = (int)(t % Wt);
  const int ty = (int)((t / Wt) % Ht);
  const int b = (int)(t / ((int64_t)Wt * Ht));
  const float* in = D + tl * C + c4 * 4;
  const int64_t pstride = Mt * C;
  f32x4 m[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) m[r][s] = *(const f32x4*)(in + (int64_t)(r * 4 + s) * pstride);
  f32x4 u[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u[0][s] = m[0][s] + m[1][s] + m[2][s];
    u[1][s] = m[1][s] - m[2][s] - m[3][s];
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *(const f32x4*)(bias + c4 * 4);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x4 o0 = u[i][0] + u[i][1] + u[i][2] + bv;
    f32x4 o1 = u[i][1] - u[i][2] - u[i][3] + bv;
    const int64_t pix = ((int64_t)b * H + 2 * ty + i) * W + 2 * tx;
    if (res) {
      o0 += *(const f32x4*)(res + pix * ldres + c4 * 4);
      o1 += *(const f32x4*)(res + (pix + 1) * ldres + c4 * 4);
    }
    *(f32x4*)(y + pix * C + c4 * 4) = o0;
    *(f32x4*)(y + (pix + 1) * C + c4 * 4) = o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1 += o0[e] + o1[e];
      s2 = fmaf(o0[e], o0[e], fmaf(o1[e], o1[e], s2));
    }
  }
  if (seg) {   // the thread's 16 values in fp32, fp64 from there on; the 4 lanes of a segment are adjacent (C4 % 4 == 0)
    double d1 = (double)s1, d2 = (double)s2;
    d1 += __shfl_xor(d1, 1, 64);
    d2 += __shfl_xor(d2, 1, 64);
    d1 += __shfl_xor(d1, 2, 64);
    d2 += __shfl_xor(d2, 2, 64);
    if ((c4 & 3) == 0) {
      double* op = seg + (((int64_t)b * (C >> 4) + (c4 >> 2)) * ((int64_t)Ht * Wt) + ((int64_t)ty * Wt + tx)) * 2;
      op[0] = d1;
      op[1] = d2;
    }
  }
}

int launch_wino_out(const float* D, const float* bias, const float* res, int ldres, float* y, int B, int H, int W,
                    int C, int64_t t0, int64_t nt, hipStream_t s, double* seg_partial) {
  KD_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "Winograd output transform needs even H, W and C % 4 == 0");
  KD_REQUIRE(!seg_partial || C % 16 == 0, "Winograd output transform: statistics need C % 16 == 0");
  KD_REQUIRE(t0 >= 0 && nt > 0 && t0 + nt <= (int64_t)B * (H / 2) * (W / 2), "tile slice out of range");
  int64_t total = nt * (C / 4);
  hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, D, bias, res, ldres, y,
                     B, H, W, C, t0, nt, seg_partial);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
