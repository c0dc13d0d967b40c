// Sampler-side kernels: x0 prediction, per-sample 0.95-quantile (radix select), the fused DDPM
// reverse step (threshold clamp + posterior mean + noise add, noise either from a tensor or
// from in-kernel Philox4x32-10), the RePaint inpaint mix / re-noise and the final clamp.
//
// Every kernel reads the current iteration index from device memory (d_iter) and looks its
// per-step scalars up in device tables, so that one captured hipGraph serves all T*R
// iterations.  iteration it -> timestep k = it / R, resample slot ri = it % R (r = R-1-ri).
#include "common.h"

#pragma clang fp contract(off)  // keep mul/add separate so that the arithmetic follows torch's op order

namespace kd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(int64_t n_items) {
  int64_t b = (n_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ------------------------------------------------------------------------- Philox4x32-10 normals
struct Philox4 {
  uint32_t v[4];
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
// four standard normals for elements 4*i4 .. 4*i4+3 of stream `sid`
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t sid, uint64_t i4) {
  Philox4 r = philox4x32_10((uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)sid, (uint32_t)(sid >> 32),
                            (uint32_t)seed, (uint32_t)(seed >> 32));
  f32x4 z;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float u1 = u01(r.v[2 * p]), u2 = u01(r.v[2 * p + 1]);
    float rad = sqrtf(-2.0f * logf(u1));
    float th = 6.283185307179586f * u2;
    z[2 * p] = rad * cosf(th);
    z[2 * p + 1] = rad * sinf(th);
  }
  return z;
}
// stream ids: purpose in the top 32 bits, iteration in the low 32
__device__ __forceinline__ uint64_t stream_id(uint32_t purpose, uint32_t it) { return ((uint64_t)purpose << 32) | it; }
enum { PURPOSE_STEP = 1, PURPOSE_INPAINT = 2, PURPOSE_RENOISE = 3, PURPOSE_USER = 16 };

__global__ void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t sid) {
  int64_t n4 = (n + 3) / 4;
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (int64_t)gridDim.x * blockDim.x) {
    f32x4 z = philox_normal4(seed, sid, (uint64_t)i4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i4 * 4 + e < n) out[i4 * 4 + e] = z[e];
  }
}
int launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t sid, hipStream_t s) {
  hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, out, n, seed, sid);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__device__ __forceinline__ f32x4 noise4(const float* noise, int64_t noise_stride, uint64_t seed, uint32_t purpose,
                                        int it, int64_t i4) {
  if (noise) return *(const f32x4*)(noise + (int64_t)it * noise_stride + i4 * 4);
  return philox_normal4(seed, stream_id(purpose, (uint32_t)it), (uint64_t)i4);
}

// ------------------------------------------------------------------------- per-step helpers
__global__ void fill_time_kernel(const float* __restrict__ table, const int* __restrict__ d_iter, int R,
                                 float* __restrict__ out, int B) {
  int i = threadIdx.x;
  if (i < B) out[i] = table[*d_iter / R];
}
int launch_fill_time(const float* table, const int* d_iter, int R, float* out, int B, hipStream_t s) {
  KD_REQUIRE(B <= 1024, "batch too large for fill_time");
  hipLaunchKernelGGL(fill_time_kernel, dim3(1), dim3(((B + 63) / 64) * 64), 0, s, table, d_iter, R, out, B);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// cond region of schedule step (*d_iter / R) out of the per-schedule table -> the live cond region (16 bytes per thread)
__global__ __launch_bounds__(256) void cond_gather_kernel(const float4* __restrict__ tab, float4* __restrict__ dst, int64_t n16,
                                                          const int* __restrict__ d_iter, int R) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = tab[(int64_t)(*d_iter / R) * n16 + i];
}
int launch_cond_gather(const void* tab, void* dst, size_t bytes, const int* d_iter, int R, hipStream_t s) {
  KD_REQUIRE(tab != nullptr && dst != nullptr, "cond gather: no table");
  KD_REQUIRE(bytes % 16 == 0 && (((uintptr_t)tab | (uintptr_t)dst) & 15) == 0, "cond gather: 16-byte granules");
  const int64_t n16 = (int64_t)(bytes / 16);
  hipLaunchKernelGGL(cond_gather_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (const float4*)tab, (float4*)dst, n16,
                     d_iter, R);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- building the per-schedule conditioning table B schedule steps at a time.  The conditioning ops are row-wise
// over the batch (time MLPs, token LayerNorm, K / V projections: sample b's rows depend on sample b's log-SNR alone), so a
// run whose B samples carry B DIFFERENT schedule steps yields in row b what a run at step k0 + b yields in every row;
// the scatter writes row b of every tensor into all B rows of table entry k0 + b.  Same kernels, same tile shapes, same
// per-row arithmetic as the in-step path: the table is bit-identical to one built step by step.
__global__ void fill_time_rows_kernel(const float* __restrict__ table, int k0, int T, float* __restrict__ out, int B) {
  const int i = threadIdx.x;
  if (i < B) out[i] = table[min(k0 + i, T - 1)];
}
int launch_fill_time_rows(const float* table, int k0, int T, float* out, int B, hipStream_t s) {
  KD_REQUIRE(B <= 1024 && T > 0, "batch too large for fill_time_rows");
  hipLaunchKernelGGL(fill_time_rows_kernel, dim3(1), dim3(((B + 63) / 64) * 64), 0, s, table, k0, T, out, B);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void cond_scatter_kernel(const float* __restrict__ ws, float* __restrict__ tab,
                                                           const CondSeg* __restrict__ segs, int nseg, uint32_t row_total, int B,
                                                           int k0, int T, int64_t cond_floats) {
  const int b = blockIdx.y, k = k0 + b;
  if (k >= T) return;
  float* row = tab + (int64_t)k * cond_floats;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < row_total; e += gridDim.x * 256) {
    int lo = 0, hi = nseg - 1;   // last segment whose start <= e
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (segs[mid].start <= e) lo = mid; else hi = mid - 1;
    }
    const CondSeg sg = segs[lo];
    const uint32_t j = e - sg.start;
    const float v = ws[(size_t)sg.off4 + (size_t)b * sg.row4 + j];
    float* dst = row + sg.off4 + j;
    for (int b2 = 0; b2 < B; ++b2) dst[(size_t)b2 * sg.row4] = v;
  }
}
int launch_cond_scatter(const float* ws, float* tab, const CondSeg* d_segs, int nseg, uint32_t row_total, int B, int k0, int T,
                        int64_t cond_floats, hipStream_t s) {
  KD_REQUIRE(ws && tab && d_segs && nseg > 0 && row_total > 0, "cond scatter: empty conditioning region");
  const unsigned gx = (unsigned)std::min<uint32_t>((row_total + 255) / 256, 1024);
  hipLaunchKernelGGL(cond_scatter_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, ws, tab, d_segs, nseg, row_total, B, k0, T,
                     cond_floats);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ void iter_inc_kernel(int* d_iter) { *d_iter += 1; }
__global__ void iter_set_kernel(int* d_iter, int v) { *d_iter = v; }
__global__ void seed_set_kernel(uint64_t* d_seed, uint64_t v) { *d_seed = v; }
int launch_seed_set(uint64_t* d_seed, uint64_t value, hipStream_t s) {
  hipLaunchKernelGGL(seed_set_kernel, dim3(1), dim3(1), 0, s, d_seed, value);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_iter_set(int* d_iter, int value, hipStream_t s) {
  hipLaunchKernelGGL(iter_set_kernel, dim3(1), dim3(1), 0, s, d_iter, value);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_iter_inc(int* d_iter, hipStream_t s) {
  hipLaunchKernelGGL(iter_inc_kernel, dim3(1), dim3(1), 0, s, d_iter);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// x0 from the model output: noise: (x - sigma*eps)/max(alpha,1e-8) ; v: alpha*x - sigma*v
__global__ void x0_kernel(const float* __restrict__ x, const float* __restrict__ pred, float* __restrict__ x0,
                          StepTables tb, const int* __restrict__ d_iter, int R, int objective, int64_t n4) {
  const int k = *d_iter / R;
  const float alpha = tb.alpha[k], sigma = tb.sigma[k];
  const float alpha_c = fmaxf(alpha, 1e-8f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 xv = *(const f32x4*)(x + i * 4), pv = *(const f32x4*)(pred + i * 4), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (objective == 0) o[e] = (xv[e] - sigma * pv[e]) / alpha_c;
      else if (objective == 1) o[e] = alpha * xv[e] - sigma * pv[e];
      else o[e] = pv[e];
    }
    *(f32x4*)(x0 + i * 4) = o;
  }
}
int launch_x0(const float* x, const float* pred, float* x0, const StepTables& tb, const int* d_iter, int R,
              int objective, int64_t n, hipStream_t s) {
  KD_REQUIRE(n % 4 == 0, "image element count must be a multiple of 4");
  hipLaunchKernelGGL(x0_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, pred, x0, tb, d_iter, R, objective,
                     n / 4);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- quantile of |x| (radix select)
// torch.quantile semantics: rank = q*(n-1) in fp32, lo = floor(rank), result = lerp(v[lo], v[lo+1], rank-lo)
// on the ascending order statistics.  |x| >= 0, so the fp32 bit pattern orders like the value:
// four 8-bit MSB-first histogram passes pin v[lo] exactly; v[lo+1] is v[lo] if the last bin holds
// more than k_rem+1 elements, else the smallest value above v[lo] (one min pass).
struct QState {
  uint32_t prefix;   // bits fixed so far (high bits)
  uint32_t k_rem;    // rank still to resolve inside the prefix bucket
  uint32_t eq_count; // after the last pass: elements equal to v[lo]
  uint32_t next_bits;// min bits strictly above v[lo]
};
// workspace layout per call: QState[B] | hist[4][B][256] (uint32)
size_t quantile_ws_bytes(int B) { return (size_t)B * sizeof(QState) + (size_t)4 * B * 256 * sizeof(uint32_t); }

__global__ void q_init_kernel(QState* st, uint32_t* hist, int B, uint32_t k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4 * B * 256) hist[i] = 0;
  if (i < B) {
    st[i].prefix = 0;
    st[i].k_rem = k;
    st[i].eq_count = 0;
    st[i].next_bits = 0xFFFFFFFFu;
  }
}

__global__ __launch_bounds__(256) void q_hist_kernel(const float* __restrict__ x, const QState* __restrict__ st,
                                                     uint32_t* __restrict__ hist, int64_t n, int pass) {
  __shared__ uint32_t h[256];
  const int b = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = st[b].prefix;
  const int shift = 24 - 8 * pass;
  const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
  const uint32_t* xb = (const uint32_t*)(x + (int64_t)b * n);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t bits = xb[i] & 0x7FFFFFFFu;
    if ((bits & himask) == prefix) atomicAdd(&h[(bits >> shift) & 255u], 1u);
  }
  __syncthreads();
  uint32_t c = h[threadIdx.x];
  if (c) atomicAdd(&hist[((int64_t)pass * gridDim.y + b) * 256 + threadIdx.x], c);
}

// The digit of this pass: the bin d with cum(d) <= k < cum(d) + h[d].  One wave per sample, 4 bins per lane and a
// shuffle prefix sum (the first form walked the 256 bins in one thread: 256 dependent loads, 18 us per pass, four
// passes per denoising step).
__global__ __launch_bounds__(64) void q_scan_kernel(QState* st, const uint32_t* __restrict__ hist, int B, int pass) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const uint32_t* h = hist + ((int64_t)pass * B + b) * 256;
  const uint32_t k = st[b].k_rem;
  uint32_t c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = h[4 * lane + j];
  const uint32_t mine = (c[0] + c[1]) + (c[2] + c[3]);
  uint32_t incl = mine;   // inclusive prefix sum over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += o;
  }
  uint32_t cum = incl - mine;   // bins before this lane's four
  // the lane that holds the digit: cum <= k < incl (exactly one lane when k < n; lane 63 otherwise, as before)
  const bool here = (k >= cum && k < incl) || (lane == 63 && k >= incl);
  if (here) {
    int d = 4 * lane;
    for (int j = 0; j < 4; ++j) {
      if (k < cum + c[j] || j == 3) {
        d = 4 * lane + j;
        break;
      }
      cum += c[j];
    }
    st[b].prefix |= (uint32_t)d << (24 - 8 * pass);
    st[b].k_rem = k - cum;
    if (pass == 3) st[b].eq_count = h[d];
  }
}

__global__ __launch_bounds__(256) void q_next_kernel(const float* __restrict__ x, QState* st, int64_t n) {
  const int b = blockIdx.y;
  const uint32_t v = st[b].prefix;
  const uint32_t* xb = (const uint32_t*)(x + (int64_t)b * n);
  uint32_t best = 0xFFFFFFFFu;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t bits = xb[i] & 0x7FFFFFFFu;
    if (bits > v && bits < best) best = bits;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)best, off, 64);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best != 0xFFFFFFFFu) atomicMin(&st[b].next_bits, best);
}

__global__ void q_final_kernel(const QState* __restrict__ st, float* __restrict__ out, int B, float w) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float lo = __uint_as_float(st[b].prefix);
  float hi = lo;
  if (w > 0.f && !(st[b].k_rem + 1 < st[b].eq_count)) hi = __uint_as_float(st[b].next_bits);
  float diff = hi - lo;
  out[b] = (w < 0.5f) ? lo + w * diff : hi - diff * (1.0f - w);  // torch lerp
}

int launch_quantile_abs(const float* x, float* out, int B, int64_t n, float q, void* ws, hipStream_t s) {
  KD_REQUIRE(n >= 1 && n < (1ll << 31), "quantile: n out of range");
  KD_REQUIRE(q >= 0.f && q <= 1.f, "quantile: q must be in [0,1]");
  QState* st = (QState*)ws;
  uint32_t* hist = (uint32_t*)((char*)ws + (size_t)B * sizeof(QState));
  float rank = q * (float)(n - 1);  // fp32, as torch computes it
  float lo_f = floorf(rank);
  uint32_t k = (uint32_t)lo_f;
  float w = rank - lo_f;
  if ((int64_t)k >= n - 1) {  // q == 1
    k = (uint32_t)(n - 1);
    w = 0.f;
  }
  int init_threads = 4 * B * 256;
  hipLaunchKernelGGL(q_init_kernel, dim3((init_threads + 255) / 256), dim3(256), 0, s, st, hist, B, k);
  int gx = (int)((n + 256 * 16 - 1) / (256 * 16));
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(q_hist_kernel, dim3(gx, B), dim3(256), 0, s, x, st, hist, n, pass);
    hipLaunchKernelGGL(q_scan_kernel, dim3(B), dim3(64), 0, s, st, hist, B, pass);
  }
  hipLaunchKernelGGL(q_next_kernel, dim3(gx, B), dim3(256), 0, s, x, st, n);
  hipLaunchKernelGGL(q_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, st, out, B, w);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- fused DDPM reverse step
// x0c = clamp(x0,-s,s)/s ; mean = alpha_next*(x*(1-c)/alpha + c*x0c) ; x = mean + noise_scale*N(0,1)
__global__ void ddpm_update_kernel(float* __restrict__ x, const float* __restrict__ x0,
                                   const float* __restrict__ s_thresh, const float* __restrict__ noise,
                                   int64_t noise_stride, const uint64_t* __restrict__ d_seed, StepTables tb,
                                   const int* __restrict__ d_iter, int R, int dynamic_threshold, int64_t per4,
                                   int64_t total4) {
  const uint64_t seed = *d_seed;  // device-resident: the captured step graph is seed-independent
  const int it = *d_iter;
  const int k = it / R;
  const float alpha = tb.alpha[k], alpha_next = tb.alpha_next[k], c = tb.c[k], ns = tb.noise_scale[k];
  const float one_minus_c = 1.0f - c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per4);
    float s = 1.0f;
    if (dynamic_threshold) s = fmaxf(s_thresh[b], 1.0f);
    f32x4 xv = *(const f32x4*)(x + i * 4), x0v = *(const f32x4*)(x0 + i * 4);
    f32x4 z = noise4(noise, noise_stride, seed, PURPOSE_STEP, it, i), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xs = fminf(fmaxf(x0v[e], -s), s) / s;
      float mean = alpha_next * (xv[e] * one_minus_c / alpha + c * xs);
      o[e] = mean + ns * z[e];
    }
    *(f32x4*)(x + i * 4) = o;
  }
}
int launch_ddpm_update(float* x, const float* x0, const float* s_thresh, const float* noise, int64_t noise_stride,
                       const uint64_t* d_seed, const StepTables& tb, const int* d_iter, int R, int dynamic_threshold, int B,
                       int64_t per, hipStream_t s) {
  KD_REQUIRE(per % 4 == 0, "per-sample element count must be a multiple of 4");
  int64_t total4 = (int64_t)B * per / 4;
  hipLaunchKernelGGL(ddpm_update_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, x0, s_thresh, noise,
                     noise_stride, d_seed, tb, d_iter, R, dynamic_threshold, per / 4, total4);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- inpainting (RePaint-style)
// x = x*(1-m) + (alpha_t*inp + sigma_t*noise)*m        (mask [B,1,HW], image [B,C,HW])
__global__ void inpaint_mix_kernel(float* __restrict__ x, const float* __restrict__ inp,
                                   const float* __restrict__ mask, const float* __restrict__ noise,
                                   int64_t noise_stride, const uint64_t* __restrict__ d_seed, StepTables tb,
                                   const int* __restrict__ d_iter, int R, int C, int64_t hw4, int64_t total4) {
  const uint64_t seed = *d_seed;
  const int it = *d_iter;
  const int k = it / R;
  const float alpha = tb.alpha[k], sigma = tb.sigma[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t plane = i / hw4, p4 = i - plane * hw4;
    int64_t b = plane / C;
    f32x4 m = *(const f32x4*)(mask + (b * hw4 + p4) * 4);
    f32x4 xv = *(const f32x4*)(x + i * 4), iv = *(const f32x4*)(inp + i * 4);
    f32x4 z = noise4(noise, noise_stride, seed, PURPOSE_INPAINT, it, i), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float noised = alpha * iv[e] + sigma * z[e];
      o[e] = m[e] != 0.f ? noised : xv[e];  // img*~mask + noised*mask with a boolean mask
    }
    *(f32x4*)(x + i * 4) = o;
  }
}
int launch_inpaint_mix(float* x, const float* inp, const float* mask, const float* noise, int64_t noise_stride,
                       const uint64_t* d_seed, const StepTables& tb, const int* d_iter, int R, int B, int C, int64_t hw,
                       hipStream_t s) {
  KD_REQUIRE(hw % 4 == 0, "H*W must be a multiple of 4");
  int64_t total4 = (int64_t)B * C * hw / 4;
  hipLaunchKernelGGL(inpaint_mix_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, inp, mask, noise,
                     noise_stride, d_seed, tb, d_iter, R, C, hw / 4, total4);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// re-noise t_next -> t after a non-final resample:  x = x*rn_a + noise*rn_b   (skipped when r == 0 or last k)
__global__ void renoise_kernel(float* __restrict__ x, const float* __restrict__ noise, int64_t noise_stride,
                               const uint64_t* __restrict__ d_seed, StepTables tb,
                               const int* __restrict__ d_iter, int R, int T, int64_t total4) {
  const uint64_t seed = *d_seed;
  const int it = *d_iter;
  const int k = it / R, ri = it - k * R;
  if (ri == R - 1 || k == T - 1) return;
  const float a = tb.rn_a[k], bco = tb.rn_b[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 xv = *(const f32x4*)(x + i * 4);
    f32x4 z = noise4(noise, noise_stride, seed, PURPOSE_RENOISE, it, i), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = xv[e] * a + z[e] * bco;
    *(f32x4*)(x + i * 4) = o;
  }
}
int launch_renoise(float* x, const float* noise, int64_t noise_stride, const uint64_t* d_seed, const StepTables& tb,
                   const int* d_iter, int R, int T, int B, int64_t per, hipStream_t s) {
  int64_t total4 = (int64_t)B * per / 4;
  hipLaunchKernelGGL(renoise_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, noise, noise_stride, d_seed, tb,
                     d_iter, R, T, total4);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// final: clamp(-1,1); paste known pixels; (x+1)/2
__global__ void finalize_kernel(float* __restrict__ x, const float* __restrict__ inp,
                                const float* __restrict__ mask, int C, int64_t hw4, int64_t total4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 xv = *(const f32x4*)(x + i * 4), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(xv[e], -1.0f), 1.0f);
    if (inp) {
      int64_t plane = i / hw4, p4 = i - plane * hw4;
      int64_t b = plane / C;
      f32x4 m = *(const f32x4*)(mask + (b * hw4 + p4) * 4), iv = *(const f32x4*)(inp + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = m[e] != 0.f ? iv[e] : o[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (o[e] + 1.0f) * 0.5f;
    *(f32x4*)(x + i * 4) = o;
  }
}
int launch_finalize(float* x, const float* inp, const float* mask, int B, int C, int64_t hw, hipStream_t s) {
  int64_t total4 = (int64_t)B * C * hw / 4;
  hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(total4)), dim3(256), 0, s, x, inp, mask, C, hw / 4, total4);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
