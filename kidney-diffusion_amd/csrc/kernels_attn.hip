// Attention core, GlobalContext pooling and the small-M ("skinny") linear layers.
//
// The attention core is < 2 % of the UNet's FLOPs (QK^T/AV 0.1-1.8 %),
// its projections are GEMMs and run on the MFMA kernel in kernels_conv.hip.  Two forms of the core,
// both a flash-style single pass with fp32 online softmax over 64-key K/V tiles staged in LDS:
// attention_mfma_kernel (launches that fill the chip) puts QK^T and PV on the fp32 matrix cores;
// attention_kernel<KS> (small launches) keeps a query row in the VGPRs of KS adjacent lanes and reads
// K/V as wave-wide broadcasts.  Multi-query layout (one shared K/V head) is handled by
// Hkv = 1.
#include "common.h"

namespace kd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float act_f(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// ------------------------------------------------------------------------- attention (D = 64)
constexpr int AT_D = 64;
constexpr int AT_KT = 64;      // keys per LDS tile
constexpr int AT_LD = AT_D + 4;  // padded K/V rows: the KS lanes of a query read KS different rows without bank conflicts

// KS = 1: one lane per query.  KS = 4 (small grids, e.g. one batch-1 patch of the ultra-res grid: 8 heads x
// 1024 queries would fill 32 workgroups): 4 adjacent lanes share a query and take every 4th key each, with
// their own online-softmax state; the states are merged with shuffles at the end.
template <int KS>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int ldq,
                                                        const float* __restrict__ null_k,
                                                        const float* __restrict__ null_v, KVSeg s0, KVSeg s1,
                                                        float* __restrict__ out, int ldo, int Nq, int Hkv,
                                                        float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[AT_KT * AT_LD];
  __shared__ __attribute__((aligned(16))) float Vs[AT_KT * AT_LD];
  const int h = blockIdx.y, b = blockIdx.z;
  const int hk = Hkv == 1 ? 0 : h;
  const int ks = threadIdx.x % KS;  // key split of this lane
  const int qi = blockIdx.x * (256 / KS) + threadIdx.x / KS;
  const bool active = qi < Nq;

  float qr[AT_D], o[AT_D];
  if (active) {
    const float* qp = q + ((int64_t)b * Nq + qi) * ldq + h * AT_D;
#pragma unroll
    for (int d4 = 0; d4 < AT_D / 4; ++d4) {
      f32x4 t = *(const f32x4*)(qp + d4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) qr[d4 * 4 + e] = t[e] * scale;
    }
  } else {
#pragma unroll
    for (int d = 0; d < AT_D; ++d) qr[d] = 0.f;
  }
#pragma unroll
  for (int d = 0; d < AT_D; ++d) o[d] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int n_null = null_k ? 1 : 0;
  const int Nk = n_null + s0.n + s1.n;
  for (int j0 = 0; j0 < Nk; j0 += AT_KT) {
    const int nj = min(AT_KT, Nk - j0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nj * (AT_D / 4); idx += 256) {
      int j = idx / (AT_D / 4), d4 = idx - j * (AT_D / 4);
      int key = j0 + j;
      const float *kp, *vp;
      if (key < n_null) {
        kp = null_k;
        vp = null_v;
      } else if (key < n_null + s0.n) {
        int64_t r = (int64_t)b * s0.n + (key - n_null);
        kp = s0.k + r * s0.ld + hk * AT_D;
        vp = s0.v + r * s0.ld + hk * AT_D;
      } else {
        int64_t r = (int64_t)b * s1.n + (key - n_null - s0.n);
        kp = s1.k + r * s1.ld + hk * AT_D;
        vp = s1.v + r * s1.ld + hk * AT_D;
      }
      *(f32x4*)(Ks + j * AT_LD + d4 * 4) = *(const f32x4*)(kp + d4 * 4);
      *(f32x4*)(Vs + j * AT_LD + d4 * 4) = *(const f32x4*)(vp + d4 * 4);
    }
    __syncthreads();
    for (int j = ks; j < nj; j += KS) {
      float s0a = 0.f, s1a = 0.f, s2a = 0.f, s3a = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < AT_D / 4; ++d4) {
        f32x4 kk = *(const f32x4*)(Ks + j * AT_LD + d4 * 4);  // one address per key split: broadcast
        s0a = fmaf(qr[d4 * 4 + 0], kk[0], s0a);
        s1a = fmaf(qr[d4 * 4 + 1], kk[1], s1a);
        s2a = fmaf(qr[d4 * 4 + 2], kk[2], s2a);
        s3a = fmaf(qr[d4 * 4 + 3], kk[3], s3a);
      }
      float sc = (s0a + s1a) + (s2a + s3a);
      float mnew = fmaxf(mrun, sc);
      float corr = expf(mrun - mnew);  // exp(-inf) = 0 on the first key
      float pj = expf(sc - mnew);
      lrun = lrun * corr + pj;
      if (corr != 1.0f) {
#pragma unroll
        for (int d = 0; d < AT_D; ++d) o[d] *= corr;
      }
#pragma unroll
      for (int d4 = 0; d4 < AT_D / 4; ++d4) {
        f32x4 vv = *(const f32x4*)(Vs + j * AT_LD + d4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[d4 * 4 + e] = fmaf(pj, vv[e], o[d4 * 4 + e]);
      }
      mrun = mnew;
    }
  }
  if (KS > 1) {  // merge the KS softmax states of a query (adjacent lanes)
    float m = mrun;
#pragma unroll
    for (int off = 1; off < KS; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float w = mrun == -INFINITY ? 0.f : expf(mrun - m);  // a split that saw no key contributes nothing
    lrun *= w;
#pragma unroll
    for (int off = 1; off < KS; off <<= 1) lrun += __shfl_xor(lrun, off, 64);
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
      float v = o[d] * w;
#pragma unroll
      for (int off = 1; off < KS; off <<= 1) v += __shfl_xor(v, off, 64);
      o[d] = v;
    }
  }
  if (active && ks == 0) {
    float inv = 1.0f / lrun;
    float* op = out + ((int64_t)b * Nq + qi) * ldo + h * AT_D;
#pragma unroll
    for (int d4 = 0; d4 < AT_D / 4; ++d4) {
      f32x4 t = {o[d4 * 4] * inv, o[d4 * 4 + 1] * inv, o[d4 * 4 + 2] * inv, o[d4 * 4 + 3] * inv};
      *(f32x4*)(op + d4 * 4) = t;
    }
  }
}

// ------------------------------------------------------------------------- attention on the matrix cores (D = 64)
// Flash-style single pass with both contractions on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation).  A wave owns 32 queries of one (batch, head); the block's 4 waves share 64-key K/V tiles in
// LDS.  The scores are computed TRANSPOSED, S^T = K Q^T (A operand = K rows from LDS, B operand = the wave's
// Q rows, held in registers for the whole pass): a lane of the 32x32 accumulator then holds 16 keys of ONE
// query, so the online softmax is in-lane but for one shuffle with the lane that holds the query's other 16
// keys, and the probabilities are already in the register layout the B operand of O^T += V^T P^T wants
// (k-step r pairs the keys (r&3) + 8(r>>2) and that + 4, one per lane half) - P never leaves the registers.
// O^T is transposed through LDS at the end for row-contiguous stores.
constexpr int AM_LD = AT_D + 4;   // padded K/V rows (ds_read_b128 of 32 different keys: 4 banks apart)

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void attention_mfma_kernel(const float* __restrict__ q, int ldq,
                                                                const float* __restrict__ null_k,
                                                                const float* __restrict__ null_v, KVSeg s0, KVSeg s1,
                                                                float* __restrict__ out, int ldo, int Nq, int Hkv,
                                                                float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float KV[2 * AT_KT * AM_LD];   // K tile | V tile; the output tile at the end
  float* Ks = KV;
  float* Vs = KV + AT_KT * AM_LD;
  const int h = blockIdx.y, b = blockIdx.z;
  const int hk = Hkv == 1 ? 0 : h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, khalf = lane >> 5;
  const int qi = blockIdx.x * 128 + wave * 32 + n;

  float qv[32];   // Q[qi][32 * khalf + j] * scale: the B operand of k-step j (dims j and j + 32)
  if (qi < Nq) {
    const float* qp = q + ((int64_t)b * Nq + qi) * ldq + h * AT_D + 32 * khalf;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      f32x4 t = *(const f32x4*)(qp + d4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) qv[d4 * 4 + e] = t[e] * scale;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) qv[j] = 0.f;
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
  float mrun = -INFINITY, lsum = 0.f;

  const int n_null = null_k ? 1 : 0;
  const int Nk = n_null + s0.n + s1.n;
  for (int j0 = 0; j0 < Nk; j0 += AT_KT) {
    const int nj = min(AT_KT, Nk - j0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < AT_KT * (AT_D / 4); idx += 256) {
      int j = idx / (AT_D / 4), d4 = idx - j * (AT_D / 4);
      f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};   // rows past the last key: zeros (their p is 0)
      if (j < nj) {
        int key = j0 + j;
        const float *kp, *vp;
        if (key < n_null) {
          kp = null_k;
          vp = null_v;
        } else if (key < n_null + s0.n) {
          int64_t r = (int64_t)b * s0.n + (key - n_null);
          kp = s0.k + r * s0.ld + hk * AT_D;
          vp = s0.v + r * s0.ld + hk * AT_D;
        } else {
          int64_t r = (int64_t)b * s1.n + (key - n_null - s0.n);
          kp = s1.k + r * s1.ld + hk * AT_D;
          vp = s1.v + r * s1.ld + hk * AT_D;
        }
        kk = *(const f32x4*)(kp + d4 * 4);
        vv = *(const f32x4*)(vp + d4 * 4);
      }
      *(f32x4*)(Ks + j * AM_LD + d4 * 4) = kk;
      *(f32x4*)(Vs + j * AM_LD + d4 * 4) = vv;
    }
    __syncthreads();

    // S^T = K Q^T for the two 32-key halves of the tile
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
      const float* kr = Ks + (kt * 32 + n) * AM_LD + 32 * khalf;
#pragma unroll
      for (int j4 = 0; j4 < 8; ++j4) {
        const f32x4 kk = *(const f32x4*)(kr + j4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[e], qv[j4 * 4 + e], st[kt], 0, 0, 0);
      }
    }
    // online softmax: this lane holds keys kt*32 + (r&3) + 8(r>>2) + 4*khalf of query n
    if (nj < AT_KT) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf >= nj) st[kt][r] = -INFINITY;
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(mrun, mloc);      // finite: every tile holds at least one key
    const float corr = expf(mrun - mnew);      // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[kt][r] = expf(st[kt][r] - mnew);
        psum += st[kt][r];
      }
    lsum = lsum * corr + psum;
    mrun = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= corr;
      o1[r] *= corr;
    }
    // O^T += V^T P^T: k-step r of half kt pairs the keys this lane half holds in register r
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vr = Vs + (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * AM_LD + n;
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], st[kt][r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], st[kt][r], o1, 0, 0, 0);
      }
  }
  const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32, 64));
  __syncthreads();   // every wave is done with the last K/V tile
  float* os = KV + (wave * 32 + n) * AM_LD;   // out tile [query][dim]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * khalf;
    os[d] = o0[r] * inv;
    os[32 + d] = o1[r] * inv;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int ql = it * 4 + (lane >> 4), d4 = lane & 15;
    const int qo = blockIdx.x * 128 + wave * 32 + ql;
    if (qo < Nq)
      *(f32x4*)(out + ((int64_t)b * Nq + qo) * ldo + h * AT_D + d4 * 4) =
          *(const f32x4*)(KV + (wave * 32 + ql) * AM_LD + d4 * 4);
  }
#endif
}

int launch_attention(const float* q, int ldq, const float* null_k, const float* null_v, KVSeg s0, KVSeg s1,
                     float* out, int ldo, int B, int Nq, int H, int Hkv, float scale, hipStream_t s) {
  KD_REQUIRE(Hkv == 1 || Hkv == H, "attention: Hkv must be 1 (multi-query) or H");
  KD_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && (s0.n == 0 || s0.ld % 4 == 0) && (s1.n == 0 || s1.ld % 4 == 0),
             "attention: strides % 4");
  KD_REQUIRE((null_k ? 1 : 0) + s0.n + s1.n > 0 && Nq > 0, "attention: empty");
  // both contractions on the matrix cores once there are 128-query blocks worth launching (a batch-1 patch with 1024
  // tokens has 64 of them: 55 us there against 174 us on the vector kernel)
  if (Nq >= 128 && (int64_t)((Nq + 127) / 128) * H * B >= 16) {
    hipLaunchKernelGGL(attention_mfma_kernel, dim3((Nq + 127) / 128, H, B), dim3(256), 0, s, q, ldq, null_k, null_v, s0,
                       s1, out, ldo, Nq, Hkv, scale);
  } else {  // small launches (batch-1 patches, the test shapes): 4 lanes per query on the vector ALU
    hipLaunchKernelGGL(attention_kernel<4>, dim3((Nq + 63) / 64, H, B), dim3(256), 0, s, q, ldq, null_k, null_v, s0, s1,
                       out, ldo, Nq, Hkv, scale);
  }
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- skinny linear (M <= 16 per pass)
// y[m][n] = act( sum_k in_act(x[m][k]) * w[n][k] + bias[n] ).  Weight-bandwidth bound: every
// wave streams 4 weight rows with 16-B loads, x is staged once per block in LDS.
constexpr int SK_M = 16;
constexpr int SK_KC = 1024;
constexpr int SK_NW = 4;  // outputs per wave

__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int ldy, int M, int K, int N, int in_act, int act) {
  __shared__ __attribute__((aligned(16))) float xs[SK_M * SK_KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = (blockIdx.x * 4 + wave) * SK_NW;
  const int m0 = blockIdx.y * SK_M;
  const int mcnt = min(SK_M, M - m0);
  float acc[SK_NW][SK_M];
#pragma unroll
  for (int i = 0; i < SK_NW; ++i)
#pragma unroll
    for (int m = 0; m < SK_M; ++m) acc[i][m] = 0.f;
  const bool vec = (K & 3) == 0 && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)w) & 15) == 0;

  for (int k0 = 0; k0 < K; k0 += SK_KC) {
    const int kc = min(SK_KC, K - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < SK_M * kc; idx += 256) {
      int m = idx / kc, kk = idx - m * kc;
      float v = m < mcnt ? x[(int64_t)(m0 + m) * ldx + k0 + kk] : 0.f;
      xs[m * SK_KC + kk] = act_f(v, in_act);
    }
    __syncthreads();
    if (vec) {
      for (int kk = lane * 4; kk < kc; kk += 256) {
        f32x4 wv[SK_NW];
#pragma unroll
        for (int i = 0; i < SK_NW; ++i) {
          int n = nb + i;
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          wv[i] = n < N ? *(const f32x4*)(w + (int64_t)n * K + k0 + kk) : z;
        }
#pragma unroll
        for (int m = 0; m < SK_M; ++m) {
          f32x4 xv = *(const f32x4*)(xs + m * SK_KC + kk);
#pragma unroll
          for (int i = 0; i < SK_NW; ++i) {
            acc[i][m] = fmaf(xv[0], wv[i][0], acc[i][m]);
            acc[i][m] = fmaf(xv[1], wv[i][1], acc[i][m]);
            acc[i][m] = fmaf(xv[2], wv[i][2], acc[i][m]);
            acc[i][m] = fmaf(xv[3], wv[i][3], acc[i][m]);
          }
        }
      }
    } else {
      for (int kk = lane; kk < kc; kk += 64) {
        float wv[SK_NW];
#pragma unroll
        for (int i = 0; i < SK_NW; ++i) wv[i] = (nb + i) < N ? w[(int64_t)(nb + i) * K + k0 + kk] : 0.f;
#pragma unroll
        for (int m = 0; m < SK_M; ++m) {
          float xv = xs[m * SK_KC + kk];
#pragma unroll
          for (int i = 0; i < SK_NW; ++i) acc[i][m] = fmaf(xv, wv[i], acc[i][m]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < SK_NW; ++i) {
#pragma unroll
    for (int m = 0; m < SK_M; ++m) {
      float t = wave_sum(acc[i][m]);
      int n = nb + i;
      if (lane == 0 && n < N && m < mcnt) {
        t += bias ? bias[n] : 0.f;
        y[(int64_t)(m0 + m) * ldy + n] = act_f(t, act);
      }
    }
  }
}

// One row of x (a batch-1 patch: GlobalContext FCs, time MLPs): a GEMV.  One wave per output row, 16-byte loads along K, x
// from LDS; N / 4 workgroups (the MFMA form below runs N / 32 workgroups of 32 rows and spends 31 of its 32 MFMA columns on
// padding: 19 us per launch for 2 MB of weights on a grid of 16 - launch-latency bound)
__global__ __launch_bounds__(256) void linear_gemv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int K, int N,
                                                          int in_act, int act) {
  extern __shared__ __attribute__((aligned(16))) float xs1[];   // K floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < K; k += 256) xs1[k] = act_f(x[k], in_act);
  __syncthreads();
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const float* wr = w + (int64_t)n * K;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int kk = lane * 4; kk < K; kk += 256) {   // (K % 4 == 0, 16-byte aligned rows: checked at launch)
    const f32x4 wv = *(const f32x4*)(wr + kk);
    const f32x4 xv = *(const f32x4*)(xs1 + kk);
    a0 = fmaf(xv[0], wv[0], a0);
    a1 = fmaf(xv[1], wv[1], a1);
    a2 = fmaf(xv[2], wv[2], a2);
    a3 = fmaf(xv[3], wv[3], a3);
  }
  float t = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) y[n] = act_f(t + (bias ? bias[n] : 0.f), act);
}

int launch_linear_gemv(const float* x, const float* w, const float* bias, float* y, int K, int N, int in_act, int act,
                       hipStream_t s) {
  KD_REQUIRE(K > 0 && N > 0 && (K & 3) == 0 && (((uintptr_t)w) & 15) == 0, "gemv: K % 4 == 0 and 16-byte aligned weights");
  hipLaunchKernelGGL(linear_gemv_kernel, dim3((N + 3) / 4), dim3(256), (size_t)K * sizeof(float), s, x, w, bias, y, K, N, in_act,
                     act);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_linear_skinny_valu(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int M,
                              int K, int N, int in_act, int act, hipStream_t s) {
  KD_REQUIRE(M > 0 && N > 0 && K > 0, "skinny linear: empty");
  dim3 grid((N + 4 * SK_NW - 1) / (4 * SK_NW), (M + SK_M - 1) / SK_M);
  hipLaunchKernelGGL(linear_skinny_kernel, grid, dim3(256), 0, s, x, ldx, w, bias, y, ldy, M, K, N, in_act, act);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- learned sinusoidal embedding
__global__ void sinu_emb_kernel(const float* __restrict__ t, const float* __restrict__ w, float* __restrict__ out,
                                int B, int half) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int W = 2 * half + 1;
  if (i >= B * W) return;
  int b = i / W, j = i - b * W;
  float tv = t[b];
  float r;
  if (j == 0) {
    r = tv;
  } else {
    int f = (j - 1) % half;
    float fr = tv * w[f] * 2.0f * 3.14159265358979323846f;  // (x*w)*2*pi, left to right as torch evaluates it
    r = (j - 1) < half ? sinf(fr) : cosf(fr);
  }
  out[i] = r;
}
int launch_sinu_emb(const float* t, const float* w, float* out, int B, int half, hipStream_t s) {
  int total = B * (2 * half + 1);
  hipLaunchKernelGGL(sinu_emb_kernel, dim3((total + 63) / 64), dim3(64), 0, s, t, w, out, B, half);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace kd
