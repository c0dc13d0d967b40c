// Wide NHWC epilogue of a 32 x 32 fp32 MFMA accumulator tile (device code, gfx950).
//
// In the C/D map of v_mfma_f32_32x32x2_f32 a lane owns ONE column (lane & 31) of 16 rows, so an epilogue straight
// from the registers moves 4 bytes per lane and instruction: 16 global stores (+ 16 or 32 loads for a residual /
// gate) per tile and wave.  Short-K layers are bound by exactly that issue rate (measured on the init convs, K = 27
// to 675: 750 of 1210 us were the epilogue; MI355X guide T21: epilogue stores are issue-bound, widen them).
// Here the wave turns its tile through a PRIVATE 4 KB LDS scratch - 16 ds_write_b32, 4 ds_read_b128, no barrier:
// only the wave's own lgkmcnt - and every lane then finishes 4 consecutive channels of 4 rows with 16-byte
// global accesses: 4 stores (+ 4 or 8 loads) per tile instead of 16 (+ 16 or 32).
#pragma once
#include "common.h"

namespace kd {

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));

struct WideEpilogue {
  float* y;               // &y[row0][col0] of the tile: row stride ldy floats; 16-B aligned, ldy % 4 == 0
  int64_t ldy;
  const float* bias;      // &bias[col0] or nullptr
  const float* res;       // &res[row0][col0] (row stride ldres) or nullptr: y += res
  int64_t ldres;
  const float* gate_src;  // &gate_src[row0][col0] (row stride ldgs) or nullptr: y += gate_src * gate[col]
  int64_t ldgs;
  const float* gate;      // &gate[col0] (per column) or nullptr
  int rows;               // valid rows of the tile (1..32)
  int cols;               // valid columns (multiple of 4, 4..32)
  int act;                // ACT_* applied to acc + bias before gate / residual
};

__device__ __forceinline__ float ep_act(float v, int act) {
  // (the hardware exp2 / reciprocal, ~1 ulp each, as the Winograd input transforms evaluate it: an IEEE division is ten
  // instructions, and the upsample convs' epilogue applies this to 64 values per lane and tile)
  if (act == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
  return v;
}

// scratch: 1024 floats of LDS owned by this wave for the duration of the call.  Every lane of the wave must call.
// STATS: returns in (s1, s2) this lane's (sum, sum of squares) over its 16 output values in fp64 - the caller
// reduces them (lanes with equal lane & 7 hold the same 4 columns; columns 0-15 live in lanes with (lane & 4) == 0).
// res_of(i, row, c4, live): the residual vector of this lane's i-th row (row = (lane >> 3) + 8 i, columns c4 .. c4 + 3).
template <bool STATS, class ResFn>
__device__ __forceinline__ void store_tile32_wide_impl(const ep_f32x16& acc, float* scratch, const WideEpilogue& e, double& s1,
                                                       double& s2, ResFn res_of) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + rbase) * 32 + col] = acc[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes have landed (no other wave reads them)
  const int rq = lane >> 3, c4 = (lane & 7) * 4;
  const bool c_ok = c4 < e.cols;
  ep_f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
  if (c_ok && e.bias) bv = *(const ep_f32x4*)(e.bias + c4);
  if (c_ok && e.gate) gv = *(const ep_f32x4*)(e.gate + c4);
  float f1 = 0.f, f2 = 0.f;   // the lane's own 16 values in fp32 (fp64 is half rate: 48 -> 31 cheaper operations), fp64 from there on
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rq + 8 * i;
    ep_f32x4 v = *(const ep_f32x4*)(scratch + row * 32 + c4);
    if (c_ok && row < e.rows) {
      v += bv;
      if (e.act != ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ep_act(v[k], e.act);
      }
      if (e.gate_src) v += *(const ep_f32x4*)(e.gate_src + row * e.ldgs + c4) * gv;
      v += res_of(i, row, c4);
      *(ep_f32x4*)(e.y + row * e.ldy + c4) = v;
      if (STATS) {
        f1 += (v[0] + v[1]) + (v[2] + v[3]);
        f2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], f2))));
      }
    }
  }
  s1 = (double)f1;
  s2 = (double)f2;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch reads done before the next tile's writes
}

template <bool STATS>
__device__ __forceinline__ void store_tile32_wide(const ep_f32x16& acc, float* scratch, const WideEpilogue& e, double& s1,
                                                  double& s2) {
  store_tile32_wide_impl<STATS>(acc, scratch, e, s1, s2, [&](int, int row, int c4) {
    const ep_f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return e.res ? *(const ep_f32x4*)(e.res + row * e.ldres + c4) : z;
  });
}

// The residual vectors of a tile requested ahead of its accumulation (the k = 15 init conv runs 345 MFMAs between the
// request and the use): rp[i] = res[row (lane >> 3) + 8 i][c4 .. c4 + 3], zeros where e.res is null or outside the tile
__device__ __forceinline__ void wide_prefetch_res(const WideEpilogue& e, ep_f32x4 (&rp)[4]) {
  const int lane = threadIdx.x & 63;
  const int rq = lane >> 3, c4 = (lane & 7) * 4;
  const ep_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rq + 8 * i;
    rp[i] = (e.res && c4 < e.cols && row < e.rows) ? *(const ep_f32x4*)(e.res + row * e.ldres + c4) : z;
  }
}
template <bool STATS>
__device__ __forceinline__ void store_tile32_wide_pre(const ep_f32x16& acc, float* scratch, const WideEpilogue& e,
                                                      const ep_f32x4 (&rp)[4], double& s1, double& s2) {
  store_tile32_wide_impl<STATS>(acc, scratch, e, s1, s2, [&](int i, int, int) { return rp[i]; });
}

// folds the (s1, s2) of store_tile32_wide over the wave: afterwards lanes 0 and 4 hold the sums of columns 0-15 and
// 16-31 of the tile
__device__ __forceinline__ void reduce_tile32_stats(double& s1, double& s2) {
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  s1 += __shfl_xor(s1, 1, 64);
  s2 += __shfl_xor(s2, 1, 64);
  s1 += __shfl_xor(s1, 2, 64);
  s2 += __shfl_xor(s2, 2, 64);
}

}  // namespace kd
