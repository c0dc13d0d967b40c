// The per-step share of the UNet's initial cross-embed convolution (CrossEmbedLayer, SURVEY A.1: three stride-1
// convs k = 3 / 7 / 15 over the network input, outputs dim/2 | dim/4 | dim/4 channels) in ONE kernel.
//
// Only x's 3 planes change between denoising steps (the cond / low-res planes' share is computed once per sampling
// call and enters as `res`, unet_build.inc), so per step this is a 3-input-channel convolution with up to 225
// taps: K = 27 / 147 / 675 against N = 64 / 32 / 32.  As three launches of the generic implicit-GEMM kernel it took
// 1.25 ms per step of the 64->256 UNet (0.31 + 0.23 + 0.72: a fourth channel of padding, K padded to 32, one
// pass over the output map and one over the residual per launch, the image packed to NHWC first).  Here:
//   * persistent workgroups (one per CU) keep ALL the weights in LDS for their whole life (dim 128: 113 KB);
//   * a workgroup takes 32 x 8 output pixels at a time: the (32+14) x (8+14) x 3 halo comes straight from the NCHW
//     planes (coalesced along x) into LDS as [row][pixel][channel], zero outside the image;
//   * for a fixed kernel row the K run (tap column, channel) is CONTIGUOUS in that layout, so the MFMA A operand
//     (v_mfma_f32_32x32x2_f32: one k per lane) is a ds_read_b32 at pixel base + immediate offset, no im2col, no
//     channel padding; runs are padded by one float to an even length (its weight is 0);
//   * wave w owns row w of the tile (32 pixels = one MFMA M tile) and all output channels: 15x23 + 7x11 + 2x3x5
//     = 452 MFMAs per tile, 3 LDS reads per MFMA; weight rows have an odd stride, pixels a stride of 3 floats:
//     both operand reads are free of bank conflicts;
//   * the epilogue adds bias or the step-invariant residual, stores NHWC (optionally into a channel slice of a
//     wider buffer) and leaves the GroupNorm partials of the output (SegSrc, common.h).
#include "common.h"
#include "epilogue.h"

#include <stdlib.h>

namespace kd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IC_TW = 32, IC_TH = 8;            // output tile
constexpr int IC_PW = IC_TW + 14, IC_PH = IC_TH + 14;   // halo patch (k = 15)
constexpr int IC_PATCH = IC_PH * IC_PW * 3 + 8;  // floats (+ zeroed tail: the padded last run reads one float past a row)
constexpr int IC_SCRATCH = 8 * 1024;             // floats: one 32 x 32 epilogue tile per wave
constexpr int IC_RUN15 = 46, IC_RUN7 = 22, IC_RUN3 = 10;          // (15|7|3) x 3 values, padded to even
constexpr int IC_K15 = 15 * IC_RUN15 + 1, IC_K7 = 7 * IC_RUN7 + 1, IC_K3 = 3 * IC_RUN3 + 1;   // odd row strides

// floats of LDS for the packed weights of (n3, n7, n15) output channels, each padded to a multiple of 32 rows
__host__ __device__ constexpr int ic_rows(int n) { return (n + 31) / 32 * 32; }
size_t init_conv_weight_floats(int n3, int n7, int n15) {
  return (size_t)ic_rows(n3) * IC_K3 + (size_t)ic_rows(n7) * IC_K7 + (size_t)ic_rows(n15) * IC_K15;
}
bool init_conv_fused_ok(int S, int n3, int n7, int n15) {
  const size_t lds = (init_conv_weight_floats(n3, n7, n15) + IC_PATCH + IC_SCRATCH) * sizeof(float);
  return S % IC_TW == 0 && S % IC_TH == 0 && lds <= 160 * 1024 && ic_rows(n3) <= 64 && ic_rows(n7) <= 32 &&
         ic_rows(n15) <= 32 && n3 % 4 == 0 && n7 % 4 == 0 && n15 % 4 == 0;
}

// OIHW [n][Itot][k][k] -> rows [n][ky][kx*3 + c] over input channels c0..c0+2, runs padded to `run`, rows to `ldk`
// (zeros), n padded to a multiple of 32 rows (zeros)
__global__ void init_conv_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int n_real, int Itot, int c0,
                                      int k, int run, int ldk, int rows) {
  const int total = rows * ldk;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int n = idx / ldk, kk = idx - n * ldk;
    const int ky = kk / run, r = kk - ky * run;
    float v = 0.f;
    if (n < n_real && ky < k && r < 3 * k) {
      const int kx = r / 3, c = r - kx * 3;
      v = w[(((int64_t)n * Itot + c0 + c) * k + ky) * k + kx];
    }
    out[idx] = v;
  }
}
int launch_init_conv_pack(const float* w3, const float* w7, const float* w15, float* out, int n3, int n7, int n15, int Itot,
                          int c0, hipStream_t s) {
  float* o3 = out;
  float* o7 = o3 + (size_t)ic_rows(n3) * IC_K3;
  float* o15 = o7 + (size_t)ic_rows(n7) * IC_K7;
  hipLaunchKernelGGL(init_conv_pack_kernel, dim3(64), dim3(256), 0, s, w3, o3, n3, Itot, c0, 3, IC_RUN3, IC_K3, ic_rows(n3));
  hipLaunchKernelGGL(init_conv_pack_kernel, dim3(64), dim3(256), 0, s, w7, o7, n7, Itot, c0, 7, IC_RUN7, IC_K7, ic_rows(n7));
  hipLaunchKernelGGL(init_conv_pack_kernel, dim3(64), dim3(256), 0, s, w15, o15, n15, Itot, c0, 15, IC_RUN15, IC_K15,
                     ic_rows(n15));
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

struct InitConvParams {
  const float* x;      // NCHW [B][3][S][S]
  const float* wp;     // init_conv_weight_floats packed weights
  const float* bias;   // [n3 + n7 + n15] or nullptr (then `res` carries it)
  const float* res;    // dense NHWC [B][S][S][n3+n7+n15] step-invariant share, or nullptr
  float* y;            // NHWC, row stride ldy, first channel at y
  double* seg;         // GroupNorm partials [B][(n3+n7+n15)/16][S*S/32][2] or nullptr
  int B, S, ldy, n3, n7, n15;
};

template <int N3T>   // 32-row tiles of the k = 3 conv (1 or 2)
__global__ __launch_bounds__(512, 1) void init_conv_kernel(InitConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r3 = N3T * 32, r7 = 32, r15 = 32;
  float* W3 = lds;
  float* W7 = W3 + r3 * IC_K3;
  float* W15 = W7 + r7 * IC_K7;
  float* patch = W15 + r15 * IC_K15;
  float* scratch = patch + IC_PATCH + (threadIdx.x >> 6) * 1024;   // 4 KB per wave (epilogue.h)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = (r3 * IC_K3 + r7 * IC_K7 + r15 * IC_K15);
  for (int i = tid; i < nw; i += 512) lds[i] = p.wp[i];
  for (int i = tid; i < 8; i += 512) patch[IC_PATCH - 8 + i] = 0.f;

  const int S = p.S, tx_n = S / IC_TW, ty_n = S / IC_TH;
  const int ntiles = p.B * tx_n * ty_n;
  const int C = p.n3 + p.n7 + p.n15;
  const int frow = lane & 31, khalf = lane >> 5;
  const int64_t plane = (int64_t)S * S;

  // halo loader: 6 values per thread, fetched into registers one tile ahead (the loads fly during the MFMAs of
  // the current tile) and written to LDS behind the barrier that ends the current tile's reads
  constexpr int NPV = (IC_PH * IC_PW * 3 + 511) / 512;
  float pv[NPV];
  auto fetch = [&](int tile) {
    const int b = tile / (tx_n * ty_n);
    const int rem = tile - b * tx_n * ty_n;
    const int y0 = (rem / tx_n) * IC_TH, x0 = (rem % tx_n) * IC_TW;
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
      const int i = tid + q * 512;
      const int c = i / (IC_PH * IC_PW), r = i - c * (IC_PH * IC_PW);
      const int py = r / IC_PW, px = r - py * IC_PW;
      const int iy = y0 - 7 + py, ix = x0 - 7 + px;
      float v = 0.f;
      if (i < IC_PH * IC_PW * 3 && iy >= 0 && iy < S && ix >= 0 && ix < S)
        v = p.x[((int64_t)b * 3 + c) * plane + (int64_t)iy * S + ix];
      pv[q] = v;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tx_n * ty_n);
    const int rem = tile - b * tx_n * ty_n;
    const int y0 = (rem / tx_n) * IC_TH, x0 = (rem % tx_n) * IC_TW;
    __syncthreads();   // the previous tile's MFMA reads of the patch are done (and the weights have landed)
#pragma unroll
    for (int q = 0; q < NPV; ++q) {
      const int i = tid + q * 512;
      if (i < IC_PH * IC_PW * 3) {
        const int c = i / (IC_PH * IC_PW), r = i - c * (IC_PH * IC_PW);
        patch[r * 3 + c] = pv[q];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);

    {   // wave w: row w of the tile (8 waves, two per SIMD); one conv after the other, each finished (stored) before
        // the next starts, so that only one set of accumulators is live
      const int row = wave;
      // epilogue: the wave turns its 32-pixel x 32-channel tile through its private LDS scratch and stores 16 B per
      // lane (epilogue.h): 4 stores + 4 residual loads per tile instead of 16 + 16
      const int64_t pix0 = ((int64_t)b * S + y0 + row) * S + x0;
      const int64_t chunk = ((int64_t)(y0 + row) * S + x0) >> 5;   // 32-pixel run index inside the image
      auto tile_of = [&](int ch0, int nreal) {   // channels ch0 + [0, 32) of the output
        WideEpilogue e;
        e.y = p.y + pix0 * p.ldy + ch0;
        e.ldy = p.ldy;
        e.bias = p.bias ? p.bias + ch0 : nullptr;
        e.res = p.res ? p.res + pix0 * C + ch0 : nullptr;
        e.ldres = C;
        e.gate_src = nullptr;
        e.ldgs = 0;
        e.gate = nullptr;
        e.rows = 32;
        e.cols = nreal < 32 ? nreal : 32;
        e.act = ACT_NONE;
        return e;
      };
      // the step-invariant share (residual) of all four tiles of this row, requested before the first MFMA: 64
      // registers that arrive under the convolutions instead of 4 dependent loads in front of every tile's stores
      ep_f32x4 rp15[4], rp7[4], rp3[N3T][4];
      wide_prefetch_res(tile_of(p.n3 + p.n7, p.n15), rp15);
      wide_prefetch_res(tile_of(p.n3, p.n7), rp7);
#pragma unroll
      for (int j = 0; j < N3T; ++j) wide_prefetch_res(tile_of(j * 32, p.n3 - j * 32), rp3[j]);
      auto finish = [&](const f32x16& acc, int ch0, int nreal, const ep_f32x4 (&rp)[4]) {
        const WideEpilogue e = tile_of(ch0, nreal);
        double s1, s2;
        if (p.seg) {   // wave-uniform
          store_tile32_wide_pre<true>(acc, scratch, e, rp, s1, s2);
          reduce_tile32_stats(s1, s2);
          const int n = ch0 + 4 * (lane & 7);   // lane 0: columns 0-15, lane 4: columns 16-31
          if ((lane & ~4) == 0 && 4 * (lane & 7) < nreal) {
            double* o = p.seg + (((int64_t)b * (C >> 4) + (n >> 4)) * (plane >> 5) + chunk) * 2;
            o[0] = s1;
            o[1] = s2;
          }
        } else {
          store_tile32_wide_pre<false>(acc, scratch, e, rp, s1, s2);
        }
      };
      // k = 15: window rows row .. row + 14, columns frow .. frow + 14.  Two accumulator chains (even / odd k pairs,
      // added at the end): hipcc sinks every operand read next to its MFMA, and with ONE chain each MFMA also
      // waits for its predecessor
      auto conv15 = [&]() {
        {
          const float* pa = patch + (row * IC_PW + frow) * 3 + khalf;
          const float* pb = W15 + frow * IC_K15 + khalf;
          f32x16 a15, b15;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            a15[r] = 0.f;
            b15[r] = 0.f;
          }
          for (int ky = 0; ky < 15; ++ky) {
#pragma unroll
            for (int kk = 0; kk + 1 < IC_RUN15 / 2; kk += 2) {
              a15 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk], pb[2 * kk], a15, 0, 0, 0);
              b15 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk + 2], pb[2 * kk + 2], b15, 0, 0, 0);
            }
            a15 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[IC_RUN15 - 2], pb[IC_RUN15 - 2], a15, 0, 0, 0);
            pa += IC_PW * 3;
            pb += IC_RUN15;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) a15[r] += b15[r];
          finish(a15, p.n3 + p.n7, p.n15, rp15);
        }
      };
      // k = 7: the centred 7 x 7 window starts 4 rows / 4 columns into the 15 x 15 one
      auto conv7 = [&]() {
        {
          const float* pa = patch + ((row + 4) * IC_PW + frow + 4) * 3 + khalf;
          const float* pb = W7 + frow * IC_K7 + khalf;
          f32x16 a7;
#pragma unroll
          for (int r = 0; r < 16; ++r) a7[r] = 0.f;
          for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kk = 0; kk < IC_RUN7 / 2; ++kk)
              a7 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk], pb[2 * kk], a7, 0, 0, 0);
            pa += IC_PW * 3;
            pb += IC_RUN7;
          }
          finish(a7, p.n3, p.n7, rp7);
        }
      };
      // k = 3: starts 6 rows / 6 columns in
      auto conv3 = [&]() {
#pragma unroll
        for (int j = 0; j < N3T; ++j) {
          const float* pa = patch + ((row + 6) * IC_PW + frow + 6) * 3 + khalf;
          const float* pb = W3 + (j * 32 + frow) * IC_K3 + khalf;
          f32x16 a3;
#pragma unroll
          for (int r = 0; r < 16; ++r) a3[r] = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kk = 0; kk < IC_RUN3 / 2; ++kk)
              a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[ky * IC_PW * 3 + 2 * kk], pb[ky * IC_RUN3 + 2 * kk], a3, 0, 0, 0);
          finish(a3, j * 32, p.n3 - j * 32, rp3[j]);
        }
      };
      // the two waves of a SIMD (w, w + 4) take the convs in opposite order: one's epilogues (LDS turn, residual
      // loads, stores, fp64 statistics) run under the other's long k = 15 MFMA chain instead of next to its epilogues
      if (wave < 4) {
        conv15();
        conv7();
        conv3();
      } else {
        conv7();
        conv3();
        conv15();
      }
    }
  }
}

int launch_init_conv(const float* x, const float* wp, const float* bias, const float* res, float* y, int ldy, double* seg,
                     int B, int S, int n3, int n7, int n15, hipStream_t s) {
  KD_REQUIRE(init_conv_fused_ok(S, n3, n7, n15), "init conv kernel: image size % 32, weights must fit LDS");
  KD_REQUIRE(ldy % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)res & 15) == 0 && ((uintptr_t)bias & 15) == 0,
             "init conv kernel: 16-byte aligned output / residual / bias rows");
  KD_REQUIRE(!seg || ((n3 % 16) == 0 && (n7 % 16) == 0 && (n15 % 16) == 0), "init conv partials: 16-channel segments");
  InitConvParams p{x, wp, bias, res, y, seg, B, S, ldy, n3, n7, n15};
  const size_t smem = (init_conv_weight_floats(n3, n7, n15) + IC_PATCH + IC_SCRATCH) * sizeof(float);
  const int ntiles = B * (S / IC_TW) * (S / IC_TH);
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    KD_HIP_CHECK(hipGetDevice(&dev));
    KD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int grid = ntiles < cus ? ntiles : cus;   // persistent: one workgroup per CU keeps the weights in LDS
  if (ic_rows(n3) == 64) {
    KD_HIP_CHECK(hipFuncSetAttribute((const void*)init_conv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(init_conv_kernel<2>, dim3(grid), dim3(512), smem, s, p);
  } else {
    KD_HIP_CHECK(hipFuncSetAttribute((const void*)init_conv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(init_conv_kernel<1>, dim3(grid), dim3(512), smem, s, p);
  }
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
