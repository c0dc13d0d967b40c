// Small helpers of the (step-invariant) text-conditioning path: token select against the learned
// null embedding, positional add, mean over tokens, classifier-free-guidance combine.
#include "common.h"

namespace kd {

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// out[b][p][:] = (p < L && mask[b][p] != 0 && !drop) ? tok[b][p][:] : null_embed[p][:]      p < P (= max_text_len)
__global__ void text_select_kernel(const float* __restrict__ tok, const float* __restrict__ mask,
                                   const float* __restrict__ null_embed, float* __restrict__ out, int L, int P, int C,
                                   int drop, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(idx % C);
    int64_t t = idx / C;
    int p = (int)(t % P);
    int64_t b = t / P;
    bool keep = !drop && p < L && mask[b * L + p] != 0.f;
    out[idx] = keep ? tok[(b * L + p) * C + c] : null_embed[(int64_t)p * C + c];
  }
}
int launch_text_select(const float* tok, const float* mask, const float* null_embed, float* out, int B, int L, int P,
                       int C, int drop, hipStream_t s) {
  int64_t total = (int64_t)B * P * C;
  hipLaunchKernelGGL(text_select_kernel, dim3(grid_for(total)), dim3(256), 0, s, tok, mask, null_embed, out, L, P, C,
                     drop, total);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// y[b][r][:] = x[b][r][:] + add[r][:]
__global__ void add_rows_bcast_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                      float* __restrict__ y, int64_t RC, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x)
    y[idx] = x[idx] + add[idx % RC];
}
int launch_add_rows_bcast(const float* x, const float* add, float* y, int B, int R, int C, hipStream_t s) {
  int64_t total = (int64_t)B * R * C;
  hipLaunchKernelGGL(add_rows_bcast_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, add, y, (int64_t)R * C, total);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// y[b][c] = mean_r x[b][r][c]
__global__ void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C, int total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += x[((int64_t)b * R + r) * C + c];
  y[i] = s / (float)R;
}
int launch_mean_rows(const float* x, float* y, int B, int R, int C, hipStream_t s) {
  int total = B * C;
  hipLaunchKernelGGL(mean_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, y, R, C, total);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// classifier-free guidance: out = null + (cond - null) * scale   (forward_with_cond_scale)
__global__ void cfg_combine_kernel(const float* __restrict__ cond, const float* __restrict__ nul,
                                   float* __restrict__ out, float scale, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = nul[i] + (cond[i] - nul[i]) * scale;
}
int launch_cfg_combine(const float* cond, const float* nul, float* out, float scale, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, s, cond, nul, out, scale, n);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// qk-norm attention variants (Unet(cosine_sim_attn=True) of imagen-pytorch 1.18.x; learned q_scale / k_scale of
// later versions): every 64-wide head segment of a row is replaced by x / max(||x||, 1e-12) (* scale_vec) -
// torch's F.normalize(dim=-1) followed by the learned per-channel scale.  One wave per (row, head), in place.
__global__ __launch_bounds__(256) void l2norm_heads_kernel(float* __restrict__ x, int ld, int64_t rows, int heads,
                                                          const float* __restrict__ scale_vec) {
  const int lane = threadIdx.x & 63;
  const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= rows * heads) return;
  float* p = x + (seg / heads) * ld + (seg % heads) * 64 + lane;
  const float v = *p;
  float ss = v * v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float r = v / fmaxf(sqrtf(ss), 1e-12f);
  *p = scale_vec ? r * scale_vec[lane] : r;
}
int launch_l2norm_heads(float* x, int ld, int64_t rows, int heads, const float* scale_vec, hipStream_t s) {
  const int64_t segs = rows * heads;
  if (segs <= 0) return 0;
  hipLaunchKernelGGL(l2norm_heads_kernel, dim3((unsigned)((segs + 3) / 4)), dim3(256), 0, s, x, ld, rows, heads, scale_vec);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
