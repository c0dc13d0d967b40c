// Final 3x3 convolution to 3 image channels (Cout = 3 is far too narrow for an MFMA tile).
//
//   out[b][o][y][x] = bias[o] + sum_{c,kh,kw} in[b][y+kh-1][x+kw-1][c] * W[o][c][kh][kw]
//
// is evaluated as (1) a 1x1 GEMM on the matrix cores, P[pixel][o*9+tap] = sum_c feat[pixel][c] *
// W[o][c][tap] (27 columns, padded to 32), and (2) the 9-tap gather-sum below, which also adds the
// step-invariant contribution of the low-res conditioning planes (`stat`, computed once per
// sampling call by final_static_kernel) and writes the NCHW prediction the sampler consumes.
#include "common.h"

namespace kd {

// w_oihw [3][Ctot][3][3] -> packed [32][C] rows n = o*9 + tap for channels [0, C); rows 27..31 zero
__global__ void pack_final_kernel(const float* __restrict__ w, float* __restrict__ out, int Ctot, int C) {
  int total = 32 * C;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int c = idx % C, n = idx / C;
    float v = 0.f;
    if (n < 27) {
      int o = n / 9, tap = n - o * 9;
      v = w[((int64_t)o * Ctot + c) * 9 + tap];
    }
    out[idx] = v;
  }
}
int launch_pack_final(const float* w, float* out, int Ctot, int C, hipStream_t s) {
  hipLaunchKernelGGL(pack_final_kernel, dim3((32 * C + 255) / 256), dim3(256), 0, s, w, out, Ctot, C);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// stat[b][o][y][x] = bias[o] + sum_{c<3,taps} lowres[b][c][y+kh-1][x+kw-1] * w[o][c0+c][kh][kw]   (NCHW in/out)
__global__ void final_static_kernel(const float* __restrict__ lowres, const float* __restrict__ w,
                                    const float* __restrict__ bias, float* __restrict__ stat, int Ctot, int c0,
                                    int H, int W, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int x = (int)(idx % W);
    int64_t t = idx / W;
    int y = (int)(t % H);
    t /= H;
    int o = (int)(t % 3);
    int64_t b = t / 3;
    float acc = bias[o];
    for (int c = 0; c < 3; ++c)
      for (int kh = 0; kh < 3; ++kh) {
        int iy = y + kh - 1;
        if (iy < 0 || iy >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
          int ix = x + kw - 1;
          if (ix < 0 || ix >= W) continue;
          acc += lowres[((b * 3 + c) * H + iy) * W + ix] * w[((int64_t)o * Ctot + c0 + c) * 9 + kh * 3 + kw];
        }
      }
    stat[idx] = acc;
  }
}
int launch_final_static(const float* lowres, const float* w, const float* bias, float* stat, int Ctot, int c0, int B,
                        int H, int W, hipStream_t s) {
  int64_t total = (int64_t)B * 3 * H * W;
  int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(final_static_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, lowres, w,
                     bias, stat, Ctot, c0, H, W, total);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// out[b][o][y][x] = (stat ? stat[b][o][y][x] : bias[o]) + sum_tap P[b][y+kh-1][x+kw-1][o*9+tap]
// One block per 4 rows x 64 pixels (a wave per row): the 6 x 66 pixel rows of P (32 floats each) are staged in LDS
// with 16-B loads - every row of P is read 1.5 times (3 times with the one-row blocks of rounds 1-3: 165 -> ~60 us at
// the headline shape) -, then every lane sums its pixel's 27 values.
constexpr int FG_ROWS = 4;
__global__ __launch_bounds__(64 * FG_ROWS) void final_gather_kernel(const float* __restrict__ P, const float* __restrict__ stat,
                                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                                   int H, int W) {
  __shared__ __attribute__((aligned(16))) float tile[FG_ROWS + 2][66][36];  // padded rows: conflict-free 4-B reads
  const int segs = (W + 63) / 64, rgs = (H + FG_ROWS - 1) / FG_ROWS;
  const int seg = blockIdx.x % segs, rg = (blockIdx.x / segs) % rgs, b = blockIdx.x / (segs * rgs);
  const int x0 = seg * 64, y0 = rg * FG_ROWS, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int idx = threadIdx.x; idx < (FG_ROWS + 2) * 66 * 8; idx += 64 * FG_ROWS) {
    int q = idx & 7, px = (idx >> 3) % 66, r = idx / (66 * 8);
    int iy = y0 + r - 1, ix = x0 + px - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const f32x4*)(P + (((int64_t)b * H + iy) * W + ix) * 32 + q * 4);
    *(f32x4*)&tile[r][px][q * 4] = v;
  }
  __syncthreads();
  const int x = x0 + lane, y = y0 + wv;
  if (x >= W || y >= H) return;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const int64_t oi = (((int64_t)b * 3 + o) * H + y) * W + x;
    float acc = stat ? stat[oi] : bias[o];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc += tile[wv + kh][lane + kw][o * 9 + kh * 3 + kw];
    out[oi] = acc;
  }
}
int launch_final_gather(const float* P, const float* stat, const float* bias, float* out, int B, int H, int W,
                        hipStream_t s) {
  const int segs = (W + 63) / 64, rgs = (H + FG_ROWS - 1) / FG_ROWS;
  hipLaunchKernelGGL(final_gather_kernel, dim3((unsigned)((int64_t)B * rgs * segs)), dim3(64 * FG_ROWS), 0, s, P, stat, bias,
                     out, H, W);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
