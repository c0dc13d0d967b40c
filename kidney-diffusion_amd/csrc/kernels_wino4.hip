// Winograd F(4x4,3x3) transforms for the deepest 3x3 convolutions of the ResnetBlocks (Cin >= 512 on 16x16 .. 32x32
// maps at the benchmark's batch).
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A     with the interpolation points 0, +-1, +-2, inf (Lavin & Gray)
//
// A 3x3 / stride 1 / pad 1 conv over [B,H,W,Cin] becomes 36 independent GEMMs D_p[t][n] = sum_c V_p[t][c] U_p[n][c], one
// per position p of the 6x6 transformed tile, over the Mt = B (H/4) (W/4) tiles of 4x4 outputs: 36 Mt Cin Cout MACs for
// 144 Mt Cin Cout of the direct convolution - 4x fewer MFMA issues, 1.78x fewer than F(2x2,3x3).  The GEMMs run on the
// buffer-DMA implicit-GEMM kernel (kernels_conv.hip, weight slab picked per tile row: ConvParams::wz_rows / wz_count);
// this file holds the HBM-bound transforms:
//
//   wino4_pack (plan build)  OIHW weights            -> U [36][Cout][Cin]
//   wino4_in   (per step)    GroupNorm+FiLM+SiLU(x)  -> V [36][Mt][Cin]     (the normalised map is never written)
//   wino4_in3  (per step)    the same, V as three bf16 planes [3][36][Cin/16][Mt][16] for the bf16x3 GEMM
//                            (kernels_gemm_bf16x3.hip: fp32-class products on the bf16 matrix pipe)
//   wino4_out  (per step)    D [36][Mt][Cout]        -> y NHWC + bias (+ residual), and the GroupNorm partial sums of y
//                                                       for the layer that normalises it next (SegSrc, common.h)
//
// V and D are 2.25x the map each (4x for F(2x2,3x3)).  Accuracy: the transform matrices have entries up to 8 and
// 1/24, so fp32 re-association costs more than in F(2x2,3x3): relative L2 against an fp64 convolution 3-4e-6 at
// Cin = 512 .. 2048 (F(2x2,3x3): 5e-7, direct fp32 chain: 2.5e-7; scratch/wino43_accuracy.py), inside the 2e-5 the
// UNet forward is held to (tests/test_fullsize_gpu.py).  The plan uses it where the 4x-smaller GEMM outweighs the
// transform passes: Cin >= 512 (Builder::wino4_ok).
#include "common.h"

namespace kd {

namespace {
// SiLU through the hardware exp2 / reciprocal (~1 ulp each), as the fused kernel evaluates it: -13 % on the input
// transform against v / (1 + expf(-v)) (round 4; the kernel is partly VALU bound)
__device__ __forceinline__ float w4_silu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// B^T applied to six values (one column or one row of the 6x6 tile)
template <class F>
__device__ __forceinline__ void w4_bt(const F (&d)[6], F (&t)[6]) {
  t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
  t[1] = (d[3] + d[4]) - 4.0f * (d[1] + d[2]);
  t[2] = (d[4] - d[3]) + 4.0f * (d[1] - d[2]);
  t[3] = (d[4] - d[2]) + 2.0f * (d[3] - d[1]);
  t[4] = (d[4] - d[2]) + 2.0f * (d[1] - d[3]);
  t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
// A^T applied to six values -> four
template <class F>
__device__ __forceinline__ void w4_at(const F (&m)[6], F (&y)[4]) {
  const F a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], d = m[3] - m[4];
  y[0] = m[0] + a + c;
  y[1] = b + 2.0f * d;
  y[2] = a + 4.0f * c;
  y[3] = b + 8.0f * d + m[5];
}
}  // namespace

__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I) {
  const int64_t total = (int64_t)O * I;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
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
      for (int s = 0; s < 6; ++s) U[(int64_t)(r * 6 + s) * total + idx] = u[s];
    }
  }
}

int launch_wino4_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s) {
  const int64_t total = (int64_t)O * I;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(wino4_pack_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, U, O, I);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// One thread: one 6x6 input tile x TWO channels (8-byte accesses); threads run along the channels (a wave reads / writes
// 512 contiguous bytes).
// PLANES: V as the three bf16 planes of the bf16x3 GEMM (kernels_gemm_bf16x3.hip), [3][36][C/16][nt][16]: a wave = 8
// consecutive tiles x the 16 channels of one k-chunk (it writes 256 consecutive bytes per position and plane, and reads
// 64 bytes per pixel and tile - the four waves of a workgroup take four consecutive chunks, 256 bytes per pixel)
template <bool PLANES>
__global__ __launch_bounds__(256) void wino4_in_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ scale_shift, int ld_ss,
                                                       float* __restrict__ V, int B, int H, int W, int C, int G, int64_t nt,
                                                       int mul_c0, float mul) {
  const int Ht = H >> 2, Wt = W >> 2, C2 = C >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nt * C2) return;   // (PLANES: whole waves, nt % 8 == 0 and C % 16 == 0)
  int c;
  int64_t t;
  if (PLANES) {
    const int64_t w = idx >> 6;
    const int lane = (int)(idx & 63), nkc = C / X3_BK;
    c = (int)(w % nkc) * X3_BK + (lane & 7) * 2;
    t = (w / nkc) * 8 + (lane >> 3);
  } else {
    c = (int)(idx % C2) * 2;
    t = idx / C2;
  }
  const int tx = (int)(t % Wt);
  const int ty = (int)((t / Wt) % Ht);
  const int b = (int)(t / ((int64_t)Wt * Ht));
  f32x2 A = {1.0f, 1.0f}, Bc = {0.0f, 0.0f};
  const bool norm = stats != nullptr;
  if (norm) {   // the folding of gn_apply_silu_kernel (kernels_norm.hip)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ce = c + e, g = ce / (C / G);
      const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
      float a = rstd * gamma[ce];
      float bb = beta[ce] - mean * a;
      if (scale_shift) {
        const float sc = scale_shift[(int64_t)b * ld_ss + ce] + 1.0f;
        const float sh = scale_shift[(int64_t)b * ld_ss + C + ce];
        a *= sc;
        bb = bb * sc + sh;
      }
      // channels >= mul_c0 hold an UNSCALED skip tensor that the layer sees times `mul` (2^-1/2): the statistics were taken
      // of the scaled tensor (SegSrc::scale), so ((mul x) - mean) rstd gamma = (mul a) x + bb
      A[e] = ce >= mul_c0 ? a * mul : a;
      Bc[e] = bb;
    }
  }
  // The 36 patch values first, unconditionally (coordinates clamped into the image; what lies outside is zeroed below, after
  // the activation - the conv pads the ACTIVATED map): 36 independent loads in flight per thread.  (Until round 5 every load
  // sat in its own bounds branch and hipcc waited for each before the next - `s_waitcnt vmcnt(0)` 36 times per thread; with
  // the loads batched the 65536-pixel x 512-channel transform went from 181 us to the ~140 its traffic takes, see
  // profiles/README.md.)
  f32x2 u[6][6];
  bool rok[6], cok[6];
  uint32_t rowp[6], colp[6];   // byte offsets (the map is below 4 GB: launch_wino4_in): one 32-bit register per load address
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int iy = 4 * ty - 1 + r;
    rok[r] = iy >= 0 && iy < H;
    rowp[r] = (uint32_t)((b * H + (iy < 0 ? 0 : iy >= H ? H - 1 : iy)) * W) * (uint32_t)(ldx * 4) + (uint32_t)(c * 4);
  }
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int ix = 4 * tx - 1 + s;
    cok[s] = ix >= 0 && ix < W;
    colp[s] = (uint32_t)(ix < 0 ? 0 : ix >= W ? W - 1 : ix) * (uint32_t)(ldx * 4);
  }
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(uint32_t)((int64_t)B * H * W * ldx * 4), 0x00020000);
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rsX, rowp[r] + colp[s], 0, 0);
      u[r][s] = f32x2{__uint_as_float(q[0]), __uint_as_float(q[1])};
    }
  // B^T d: column by column, in place
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    f32x2 d[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      f32x2 v = u[r][s];
      if (norm) {
        v[0] = w4_silu(v[0] * A[0] + Bc[0]);
        v[1] = w4_silu(v[1] * A[1] + Bc[1]);
      }
      const f32x2 z = {0.0f, 0.0f};
      d[r] = (rok[r] && cok[s]) ? v : z;
    }
    f32x2 tcol[6];
    w4_bt(d, tcol);
#pragma unroll
    for (int r = 0; r < 6; ++r) u[r][s] = tcol[r];
  }
  if (PLANES) {
    // dword (two bf16) units: position stride nt C / 2, plane stride 36 of them
    const int64_t pstride = nt * C2, plane = 36 * pstride;
    uint32_t* out = (uint32_t*)V + ((int64_t)(c / X3_BK) * nt + t) * (X3_BK / 2) + (c % X3_BK) / 2;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      f32x2 trow[6];
      w4_bt(u[r], trow);
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        uint32_t h, m, l;
        x3_split(trow[s], h, m, l);
        uint32_t* o = out + (int64_t)(r * 6 + s) * pstride;
        o[0] = h;
        o[plane] = m;
        o[2 * plane] = l;
      }
    }
    return;
  }
  float* out = V + t * C + c;
  const int64_t pstride = nt * C;
#pragma unroll
  for (int r = 0; r < 6; ++r) {   // (B^T d) B: row by row
    f32x2 trow[6];
    w4_bt(u[r], trow);
#pragma unroll
    for (int s = 0; s < 6; ++s) *(f32x2*)(out + (int64_t)(r * 6 + s) * pstride) = trow[s];
  }
}

// (Round 4, measured and removed, plane form: the four lanes of a quad transposing four positions' dwords (DPP quad_perm) so
// that every lane stores 16 bytes - whole 128-byte lines per wave instruction instead of 32 bytes per tile: 186.6 us against
// 186.0 on 65536 pixels x 512 channels.  The kernel is bound by the WRITE rate of the memory side: 442 MB written + 185 MB read
// in 186 us, and its fp32 form 302 MB written in 129 us - 2.3-2.4 TB/s of writes either way, while the read-mostly kernels of
// the engine (gate_add, wino4_out) move 5.5-6.0 TB/s.)
// (Round 4, measured and removed: a 16-byte variant - one thread = tile x FOUR channels x one half of the transformed rows,
// 30 of the 36 pixels each - took 1715 us per step over the 31 launches against 1369 us of this kernel: the width of the
// accesses is not what limits it.)
int launch_wino4_in(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                    const float* scale_shift, int ld_ss, float* V, int B, int H, int W, int C, int G, hipStream_t s, int mul_c0,
                    float mul) {
  KD_REQUIRE(mul_c0 < 0 || stats, "Winograd F(4x4,3x3) input transform: a channel scale needs the GroupNorm form");
  if (mul_c0 < 0) mul_c0 = C;
  KD_REQUIRE(H % 4 == 0 && W % 4 == 0 && ldx >= C && C % 2 == 0 && ldx % 2 == 0 && ((uintptr_t)x & 7) == 0,
             "Winograd F(4x4,3x3) input transform needs H % 4 == 0, W % 4 == 0 and even C / row stride");
  KD_REQUIRE((int64_t)B * H * W * ldx * 4 < ((int64_t)1 << 32), "Winograd F(4x4,3x3) input transform: maps below 4 GB");
  const int64_t nt = (int64_t)B * (H / 4) * (W / 4), total = nt * (C / 2);
  hipLaunchKernelGGL(wino4_in_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, stats, gamma,
                     beta, scale_shift, ld_ss, V, B, H, W, C, G, nt, mul_c0, mul);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_wino4_in3(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                     const float* scale_shift, int ld_ss, void* V3, int B, int H, int W, int C, int G, hipStream_t s, int mul_c0,
                     float mul) {
  KD_REQUIRE(mul_c0 < 0 || stats, "Winograd F(4x4,3x3) input transform: a channel scale needs the GroupNorm form");
  if (mul_c0 < 0) mul_c0 = C;
  const int64_t nt = (int64_t)B * (H / 4) * (W / 4), total = nt * (C / 2);
  KD_REQUIRE((int64_t)B * H * W * ldx * 4 < ((int64_t)1 << 32), "Winograd F(4x4,3x3) input transform: maps below 4 GB");
  KD_REQUIRE(H % 4 == 0 && W % 4 == 0 && ldx >= C && C % X3_BK == 0 && ldx % 2 == 0 && ((uintptr_t)x & 7) == 0 && nt % 8 == 0 &&
                 ((uintptr_t)V3 & 3) == 0,
             "Winograd F(4x4,3x3) input transform to bf16x3 planes needs H % 4 == 0, W % 4 == 0, C % 16 == 0, tiles % 8 == 0");
  hipLaunchKernelGGL(wino4_in_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, stats, gamma, beta,
                     scale_shift, ld_ss, (float*)V3, B, H, W, C, G, nt, mul_c0, mul);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// One thread: one tile x TWO channels -> 4x4 outputs (8-byte accesses); a wave = one tile x 128 consecutive channels
// (C % 128 == 0) or a narrower run.  seg != nullptr: fp64 (sum, sum of squares) of the 16 outputs x 16 channels of every
// 16-channel segment, entry [b][c / 16][tile of the image][2] (one chunk per tile: nchunk = (H/4)(W/4))
template <bool RES>   // (its own kernel: the residual's registers cost the plain form a wave per SIMD)
__global__ __launch_bounds__(256) void wino4_out_kernel(const float* __restrict__ D, const float* __restrict__ bias,
                                                        const float* __restrict__ res, int ldres, float* __restrict__ y,
                                                        int ldy, double* __restrict__ seg, int B, int H, int W, int C,
                                                        int64_t nt) {
  const int Ht = H >> 2, Wt = W >> 2, C2 = C >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nt * C2) return;   // (whole waves: nt C / 2 % 64 == 0 is required when seg != nullptr)
  const int c = (int)(idx % C2) * 2;
  const int64_t t = idx / C2;
  const int tx = (int)(t % Wt);
  const int ty = (int)((t / Wt) % Ht);
  const int b = (int)(t / ((int64_t)Wt * Ht));
  const float* in = D + t * C + c;
  const int64_t pstride = nt * C;
  f32x2 u[4][6];   // A^T m: column by column
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    f32x2 m[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) m[r] = *(const f32x2*)(in + (int64_t)(r * 6 + s) * pstride);
    f32x2 ycol[4];
    w4_at(m, ycol);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i][s] = ycol[i];
  }
  f32x2 bv = {0.0f, 0.0f};
  if (bias) bv = *(const f32x2*)(bias + c);
  // the residual a ROW of four pixels at a time, row i + 1's loads in flight under row i's transform and stores (round 5: as
  // `if (res) v += ...` inside the loop each of the 16 loads had a full wait behind it; all 16 together with the 36 D values
  // cost a wave per SIMD)
  f32x2 rnext[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rnext[j] = f32x2{0.0f, 0.0f};
  const int64_t pix0 = ((int64_t)b * H + 4 * ty) * W + 4 * tx;
  if (RES) {
#pragma unroll
    for (int j = 0; j < 4; ++j) rnext[j] = *(const f32x2*)(res + (pix0 + j) * ldres + c);
  }
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 rcur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rcur[j] = rnext[j];
    if (RES && i < 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rnext[j] = *(const f32x2*)(res + (pix0 + (int64_t)(i + 1) * W + j) * ldres + c);
    }
    f32x2 o[4];
    w4_at(u[i], o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t pix = pix0 + (int64_t)i * W + j;
      f32x2 v = o[j] + bv;
      v += rcur[j];
      *(f32x2*)(y + pix * ldy + c) = v;
      s1 += v[0] + v[1];
      s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], s2));
    }
  }
  if (seg) {   // the thread's 32 values in fp32, fp64 from there on; the 8 lanes of a 16-channel segment are adjacent
    double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
      d1 += __shfl_xor(d1, off, 64);
      d2 += __shfl_xor(d2, off, 64);
    }
    if ((c & 15) == 0) {
      const int nchunk = Ht * Wt;
      double* op = seg + (((int64_t)b * (C >> 4) + (c >> 4)) * nchunk + (ty * Wt + tx)) * 2;
      op[0] = d1;
      op[1] = d2;
    }
  }
}

int launch_wino4_out(const float* D, const float* bias, const float* res, int ldres, float* y, int ldy, double* seg_partial,
                     int B, int H, int W, int C, hipStream_t s) {
  KD_REQUIRE(H % 4 == 0 && W % 4 == 0 && ldy >= C && C % 2 == 0 && ldy % 2 == 0 && ldres % 2 == 0 &&
                 (((uintptr_t)y | (uintptr_t)res | (uintptr_t)bias) & 7) == 0,
             "Winograd F(4x4,3x3) output transform needs H % 4 == 0, W % 4 == 0, even C / row strides, 8-byte aligned maps");
  KD_REQUIRE(!seg_partial || C % 128 == 0 || (C % 16 == 0 && ((int64_t)B * (H / 4) * (W / 4) * (C / 2)) % 64 == 0),
             "output statistics need whole waves of 16-channel segments");
  const int64_t nt = (int64_t)B * (H / 4) * (W / 4), total = nt * (C / 2);
  if (res)
    hipLaunchKernelGGL(wino4_out_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, D, bias, res, ldres, y, ldy,
                       seg_partial, B, H, W, C, nt);
  else
    hipLaunchKernelGGL(wino4_out_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, D, bias, res, ldres, y, ldy,
                       seg_partial, B, H, W, C, nt);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
