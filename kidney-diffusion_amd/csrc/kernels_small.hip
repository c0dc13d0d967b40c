// Small-M linear layers on the matrix cores and the single-pass GlobalContext pooling.
#include "common.h"

namespace kd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_s(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------- skinny linear, M <= 32 per pass
// y[m][n] = act( sum_k in_act(x[m][k]) * w[n][k] + bias[n] ).  These layers (time MLPs: 113 MB of
// weights per step; GlobalContext FCs; time-token projections) are weight-bandwidth bound: each
// wave owns 32 output columns and streams their weight rows straight from HBM into MFMA A-operand
// registers (lane (i,h) holds w[n0+i][k+4h..k+4h+3]); x (<= 32 rows, L2-resident) is the B operand.
// v_mfma_f32_32x32x2_f32 keeps the arithmetic exact fp32 and leaves the VALU to the activation.
constexpr int SKM_UNROLL = 8;

// KW waves per workgroup split K between them (wave w takes a contiguous K range) and wave 0 adds the partial
// tiles through LDS in wave order: 4x the loads in flight for the same 32 output columns.  A batch-1 patch streams
// 228 MB of time-MLP weights through N/32 = 1744 single-wave workgroups otherwise (2 TB/s), and a GlobalContext FC
// of 2048 -> 1024 through 32 of them (0.13 TB/s).
template <int KW>
__global__ __launch_bounds__(64 * KW) void linear_skinny_mfma_kernel(const float* __restrict__ x, int ldx,
                                                                    const float* __restrict__ w,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ y, int ldy, int M, int K, int N,
                                                                    int in_act, int act) {
  __shared__ float part[KW > 1 ? (KW - 1) * 16 * 64 : 1];
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int kw = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int n = min(n0 + i, N - 1);  // clamped: rows past N are computed and dropped
  const int m = m0 + i;
  const bool m_ok = m < M;
  const float* wp = w + (int64_t)n * K + 4 * h;
  const float* xp = x + (int64_t)(m_ok ? m : 0) * ldx + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const int kper = KW > 1 ? ((K + KW * 8 * SKM_UNROLL - 1) / (KW * 8 * SKM_UNROLL)) * (8 * SKM_UNROLL) : K;
  const int k_lo = kw * kper, k_hi = min(K, k_lo + kper);
  for (int k0 = k_lo; k0 < k_hi; k0 += 8 * SKM_UNROLL) {
    f32x4 a[SKM_UNROLL], b[SKM_UNROLL];
#pragma unroll
    for (int u = 0; u < SKM_UNROLL; ++u) {
      int k = k0 + 8 * u + 4 * h;
      bool ok = k < k_hi;  // K % 4 == 0 and kper % 64 == 0, so a 4-wide slice is either fully inside or outside
      a[u] = ok ? *(const f32x4*)(wp + k0 + 8 * u) : z;
      b[u] = (ok && m_ok) ? *(const f32x4*)(xp + k0 + 8 * u) : z;
    }
#pragma unroll
    for (int u = 0; u < SKM_UNROLL; ++u) {
      if (in_act != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = act_s(b[u][e], in_act);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s], b[u][s], acc, 0, 0, 0);
    }
  }
  if (KW > 1) {   // fixed-order sum of the K ranges
    if (kw > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((kw - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kw > 0) return;
#pragma unroll
    for (int q = 0; q < KW - 1; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += part[(q * 16 + r) * 64 + lane];
  }
  // D[i=n][j=m]: col = lane&31 = m, row = (r&3) + 8*(r>>2) + 4*h
  if (!m_ok) return;
  float* yr = y + (int64_t)m * ldy;
  const bool vec = (ldy & 3) == 0 && (((uintptr_t)y) & 15) == 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nb = n0 + 8 * g + 4 * h;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int nn = nb + e;
      v[e] = act_s(acc[4 * g + e] + ((bias && nn < N) ? bias[nn] : 0.f), act);
    }
    if (vec && nb + 3 < N) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *(f32x4*)(yr + nb) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nb + e < N) yr[nb + e] = v[e];
    }
  }
}

int launch_linear_skinny(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int M,
                         int K, int N, int in_act, int act, hipStream_t s) {
  KD_REQUIRE(M > 0 && N > 0 && K > 0, "skinny linear: empty");
  const bool mfma_ok = (K & 3) == 0 && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)w) & 15) == 0;
  if (!mfma_ok) return launch_linear_skinny_valu(x, ldx, w, bias, y, ldy, M, K, N, in_act, act, s);
  if (M == 1 && K <= 16384) return launch_linear_gemv(x, w, bias, y, K, N, in_act, act, s);
  dim3 grid((N + 31) / 32, (M + 31) / 32);
  if (K >= 512)
    hipLaunchKernelGGL(linear_skinny_mfma_kernel<4>, grid, dim3(256), 0, s, x, ldx, w, bias, y, ldy, M, K, N, in_act, act);
  else
    hipLaunchKernelGGL(linear_skinny_mfma_kernel<1>, grid, dim3(64), 0, s, x, ldx, w, bias, y, ldy, M, K, N, in_act, act);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- GlobalContext pooling, single pass
// logits[p] = x[p][:]·wk + bk ; pooled[c] = sum_p softmax_p(logits) x[p][c]   (per batch element)
// One read of x: every wave keeps a running (max, sum, weighted channel sums) over its rows (online
// softmax), waves are merged through LDS, blocks through a small partial buffer.
// pixels per block: 256, more where an image would otherwise have more than 1024 blocks, fewer (down to 32) where the
// launch would otherwise have fewer than 1024 blocks - the 16x16 / 32x32 levels and every level of a batch-1 patch
// ran on 16 .. 256 blocks (a batch-1 64x64 x 1024-channel map: 43 us for 17 MB)
constexpr int GCA_ROWS = 256;
static inline int gca_rows(int HW, int B) {
  int r = GCA_ROWS;
  while ((HW + r - 1) / r > 1024) r *= 2;
  while (r > 32 && (int64_t)((HW + r - 1) / r) * B < 1024) r /= 2;
  return r;
}
constexpr int GCA_MAXT = 8;     // float4 slices per lane: C <= 2048
constexpr int GCA_U = 4;        // rows in flight per wave

// T float4 slices per lane; RPW rows per wave-instruction (2 when a row is only 32 float4 wide)
template <int T, int RPW>
__global__ __launch_bounds__(256) void gca_partial_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                          const float* __restrict__ bk, float* __restrict__ part,
                                                          int HW, int C, int rows) {
  extern __shared__ float sm[];  // [4][C] + [4][2]
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = 64 / RPW;
  const int sub = RPW == 2 ? (lane >> 5) : 0;
  const int li = RPW == 2 ? (lane & 31) : lane;
  const int C4 = C >> 2;
  const int p0 = chunk * rows, p1 = min(HW, p0 + rows);
  f32x4 wv[T], acc[T];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int c4 = li + LPR * t;
    wv[t] = c4 < C4 ? *(const f32x4*)(wk + c4 * 4) : z;
    acc[t] = z;
  }
  const float bias = bk[0];
  float mrun = -INFINITY, lrun = 0.f;
  const int stride = 4 * RPW;
  for (int pb = p0 + wave * RPW + sub; pb < p1; pb += stride * GCA_U) {
    f32x4 xv[GCA_U][T];
#pragma unroll
    for (int u = 0; u < GCA_U; ++u) {
      const int p = pb + u * stride;
      const float* xr = x + ((int64_t)b * HW + (p < p1 ? p : p0)) * C;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        int c4 = li + LPR * t;
        xv[u][t] = c4 < C4 ? *(const f32x4*)(xr + c4 * 4) : z;
      }
    }
#pragma unroll
    for (int u = 0; u < GCA_U; ++u) {
      float d = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t)
        d += (xv[u][t][0] * wv[t][0] + xv[u][t][1] * wv[t][1]) + (xv[u][t][2] * wv[t][2] + xv[u][t][3] * wv[t][3]);
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      if (pb + u * stride < p1) {  // uniform within the row's lanes
        const float logit = d + bias;
        const float mnew = fmaxf(mrun, logit);
        const float corr = expf(mrun - mnew), pw = expf(logit - mnew);
        lrun = lrun * corr + pw;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = acc[t] * corr + xv[u][t] * pw;
        mrun = mnew;
      }
    }
  }
  if (RPW == 2) {  // merge the two half-wave states
    const float mo = __shfl_xor(mrun, 32, 64), lo = __shfl_xor(lrun, 32, 64);
    const float mn = fmaxf(mrun, mo);
    const float e1 = mrun == -INFINITY ? 0.f : expf(mrun - mn), e2 = mo == -INFINITY ? 0.f : expf(mo - mn);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 ao;
#pragma unroll
      for (int e = 0; e < 4; ++e) ao[e] = __shfl_xor(acc[t][e], 32, 64);
      acc[t] = acc[t] * e1 + ao * e2;
    }
    lrun = lrun * e1 + lo * e2;
    mrun = mn;
  }
  // merge the four waves
  float* sacc = sm;          // [4][C]
  float* sml = sm + 4 * C;   // [4][2]
  if (lane == 0) {
    sml[wave * 2] = mrun;
    sml[wave * 2 + 1] = lrun;
  }
  if (sub == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int c4 = li + LPR * t;
      if (c4 < C4) *(f32x4*)(sacc + wave * C + c4 * 4) = acc[t];
    }
  }
  __syncthreads();
  float mb = fmaxf(fmaxf(sml[0], sml[2]), fmaxf(sml[4], sml[6]));
  float sc[4], lb = 0.f;
#pragma unroll
  for (int wv_ = 0; wv_ < 4; ++wv_) {
    sc[wv_] = sml[wv_ * 2] == -INFINITY ? 0.f : expf(sml[wv_ * 2] - mb);
    lb += sml[wv_ * 2 + 1] * sc[wv_];
  }
  float* out = part + ((int64_t)b * nchunks + chunk) * (C + 2);
  for (int c = threadIdx.x; c < C; c += 256)
    out[2 + c] = (sacc[c] * sc[0] + sacc[C + c] * sc[1]) + (sacc[2 * C + c] * sc[2] + sacc[3 * C + c] * sc[3]);
  if (threadIdx.x == 0) {
    out[0] = mb;
    out[1] = lb;
  }
}

__global__ __launch_bounds__(256) void gca_combine_kernel(const float* __restrict__ part, float* __restrict__ pooled,
                                                          int nchunks, int C) {
  // grid (B, ceil(C/16)): a block merges 16 channels over all chunks - lane & 15 = channel, the other 4 thread-index
  // bits = one of 16 chunk subsets (i = sub, sub + 16, ..).  (Round 2 ran 64 channels per block with 4 chunk subsets: a
  // batch-1 patch with 1024 chunks then had C / 64 = 2 .. 16 blocks walking 256 dependent loads each - 40 of the 56 us
  // of a stage-3 GlobalContext pooling.)  Fixed summation order: bit-identical from run to run.
  __shared__ float se[1024];   // exp(m_i - mg) per chunk (nchunks <= 1024)
  __shared__ float red[8];
  __shared__ float sacc[16][16];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int c = blockIdx.y * 16 + cl;
  const float* pb = part + (int64_t)b * nchunks * (C + 2);
  float m = -INFINITY;
  for (int i = threadIdx.x; i < nchunks; i += 256) m = fmaxf(m, pb[(int64_t)i * (C + 2)]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  const float mg = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 256) {
    float e = expf(pb[(int64_t)i * (C + 2)] - mg);
    se[i] = e;
    l += pb[(int64_t)i * (C + 2) + 1] * e;
  }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  float s = 0.f;
  if (c < C)
    for (int i = sub; i < nchunks; i += 16) s += pb[(int64_t)i * (C + 2) + 2 + c] * se[i];
  sacc[sub][cl] = s;
  __syncthreads();
  if (sub == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sacc[k][cl];
    pooled[(int64_t)b * C + c] = t * inv;
  }
}

// ------------------------------------------------------------------------- GlobalContext gate in one launch (round 4)
// The merge of the pooling partials, FC (C -> hid) + SiLU and FC (hid -> C) + sigmoid for one image per workgroup: three
// launches (gca_combine + two skinny linears: 5 + 13-20 + 13-20 us of dependent-load latency) become one.  For
// C <= 512: a workgroup then streams at most 0.5 + 0.5 MB of weights; above that the skinny kernels' N / 32 workgroups
// stream them faster than 16 workgroups can.  Plain fp32 fma chains in index order.
constexpr int GCAG_MAXC = 512;
__global__ __launch_bounds__(256) void gca_gate_kernel(const float* __restrict__ part, int nchunks, int C,
                                                       const float* __restrict__ w0, const float* __restrict__ b0, int hid,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ gate) {
  __shared__ float se[1024];   // exp(m_i - mg) per chunk (nchunks <= 1024)
  __shared__ float red[8];
  __shared__ __attribute__((aligned(16))) float pooled[GCAG_MAXC];
  __shared__ __attribute__((aligned(16))) float hidden[GCAG_MAXC / 2];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pb = part + (int64_t)b * nchunks * (C + 2);
  float m = -INFINITY;
  for (int i = threadIdx.x; i < nchunks; i += 256) m = fmaxf(m, pb[(int64_t)i * (C + 2)]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  const float mg = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 256) {
    const float e = expf(pb[(int64_t)i * (C + 2)] - mg);
    se[i] = e;
    l += pb[(int64_t)i * (C + 2) + 1] * e;
  }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
  for (int c = threadIdx.x; c < C; c += 256) {   // consecutive threads read consecutive channels of a chunk
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 3 < nchunks; i += 4) {
      s0 = fmaf(pb[(int64_t)i * (C + 2) + 2 + c], se[i], s0);
      s1 = fmaf(pb[(int64_t)(i + 1) * (C + 2) + 2 + c], se[i + 1], s1);
      s2 = fmaf(pb[(int64_t)(i + 2) * (C + 2) + 2 + c], se[i + 2], s2);
      s3 = fmaf(pb[(int64_t)(i + 3) * (C + 2) + 2 + c], se[i + 3], s3);
    }
    for (; i < nchunks; ++i) s0 = fmaf(pb[(int64_t)i * (C + 2) + 2 + c], se[i], s0);
    pooled[c] = ((s0 + s1) + (s2 + s3)) * inv;
  }
  __syncthreads();
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  for (int h = threadIdx.x; h < hid; h += 256) {   // hidden = SiLU(W0 pooled + b0): a thread streams one weight row
    const f32x4g* wr = (const f32x4g*)(w0 + (int64_t)h * C);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const f32x4g wv = wr[c4], pv = *(const f32x4g*)(pooled + 4 * c4);
      a0 = fmaf(wv[0], pv[0], a0);
      a1 = fmaf(wv[1], pv[1], a1);
      a2 = fmaf(wv[2], pv[2], a2);
      a3 = fmaf(wv[3], pv[3], a3);
    }
    const float v = ((a0 + a1) + (a2 + a3)) + b0[h];
    hidden[h] = v / (1.0f + expf(-v));
  }
  __syncthreads();
  for (int n = threadIdx.x; n < C; n += 256) {     // gate = sigmoid(W2 hidden + b2)
    const float* wr = w2 + (int64_t)n * hid;
    float a0 = 0.f, a1 = 0.f;
    int h = 0;
    if ((hid & 3) == 0) {
      float a2 = 0.f, a3 = 0.f;
      for (; h < hid; h += 4) {
        const f32x4g wv = *(const f32x4g*)(wr + h), hv = *(const f32x4g*)(hidden + h);
        a0 = fmaf(wv[0], hv[0], a0);
        a1 = fmaf(wv[1], hv[1], a1);
        a2 = fmaf(wv[2], hv[2], a2);
        a3 = fmaf(wv[3], hv[3], a3);
      }
      a0 += a2;
      a1 += a3;
    } else {
      for (; h < hid; ++h) a0 = fmaf(wr[h], hidden[h], a0);
    }
    const float v = (a0 + a1) + b2[n];
    gate[(int64_t)b * C + n] = 1.0f / (1.0f + expf(-v));
  }
}

bool gca_gate_fused_ok(int C, int hid) {
  return C % 4 == 0 && C <= GCAG_MAXC && hid <= GCAG_MAXC / 2 && hid > 0;
}

static int launch_gca_partial(const float* x, const float* wk, const float* bk, float* scratch, int B, int HW, int C, int rows,
                              int chunks, hipStream_t s) {
  size_t smem = (size_t)(4 * C + 8) * sizeof(float);
  const int C4 = C / 4;
  dim3 grid(chunks, B), blk(256);
  if (C4 == 32) hipLaunchKernelGGL((gca_partial_kernel<1, 2>), grid, blk, smem, s, x, wk, bk, scratch, HW, C, rows);
  else if (C4 <= 64) hipLaunchKernelGGL((gca_partial_kernel<1, 1>), grid, blk, smem, s, x, wk, bk, scratch, HW, C, rows);
  else if (C4 <= 128) hipLaunchKernelGGL((gca_partial_kernel<2, 1>), grid, blk, smem, s, x, wk, bk, scratch, HW, C, rows);
  else if (C4 <= 256) hipLaunchKernelGGL((gca_partial_kernel<4, 1>), grid, blk, smem, s, x, wk, bk, scratch, HW, C, rows);
  else hipLaunchKernelGGL((gca_partial_kernel<8, 1>), grid, blk, smem, s, x, wk, bk, scratch, HW, C, rows);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// pooling partials + the gate in two launches: gate[b][c] = sigmoid(W2 SiLU(W0 pooled_b + b0) + b2)
int launch_gca_gate(const float* x, const float* wk, const float* bk, float* scratch, const float* w0, const float* b0, int hid,
                    const float* w2, const float* b2, float* gate, int B, int HW, int C, hipStream_t s) {
  KD_REQUIRE(gca_gate_fused_ok(C, hid), "fused GlobalContext gate needs C % 4 == 0, C <= 512, hidden <= 256");
  KD_REQUIRE((((uintptr_t)w0 | (uintptr_t)w2) & 15) == 0, "fused GlobalContext gate: 16-byte aligned weights");
  const int rows = gca_rows(HW, B), chunks = (HW + rows - 1) / rows;
  if (launch_gca_partial(x, wk, bk, scratch, B, HW, C, rows, chunks, s)) return 1;
  hipLaunchKernelGGL(gca_gate_kernel, dim3(B), dim3(256), 0, s, scratch, chunks, C, w0, b0, hid, w2, b2, gate);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t gca_scratch_floats(int B, int HW, int C) {
  int rows = gca_rows(HW, B), chunks = (HW + rows - 1) / rows;
  return (size_t)B * chunks * (C + 2);
}

int launch_gca_pool(const float* x, const float* wk, const float* bk, float* /*logits (unused)*/, float* pooled,
                    float* scratch, int B, int HW, int C, hipStream_t s) {
  KD_REQUIRE(C % 4 == 0 && C <= 64 * 4 * GCA_MAXT, "gca needs C % 4 == 0 and C <= 2048");
  const int rows = gca_rows(HW, B), chunks = (HW + rows - 1) / rows;
  if (launch_gca_partial(x, wk, bk, scratch, B, HW, C, rows, chunks, s)) return 1;
  hipLaunchKernelGGL(gca_combine_kernel, dim3(B, (C + 15) / 16), dim3(256), 0, s, scratch, pooled, chunks, C);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
