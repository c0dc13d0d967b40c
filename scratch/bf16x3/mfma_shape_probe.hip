// EXPERIMENT: does the six-product bf16x3 inner loop run faster on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16?
// (MI355X_MICROARCH.md: bare loops on random data, operands re-read from LDS: 1.12-1.14 x the FLOP/s at equal cycles per FLOP -
// a clock effect.)  Operands sit in LDS (random bf16 bits of moderate exponent), no global traffic in the loop: every wave
// computes a 64 x 64 tile, 32 k per iteration, six products, fragments re-read from LDS each iteration - the instruction mix of
// gemm_bf16x3_kernel's computing waves.  8 waves per workgroup (2 per SIMD), one workgroup per CU.
//
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_shape_probe.hip -o libmfma_probe.so ; run: mfma_shape_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 0, 1, 0, 1, 2};

// LDS image: A [4 row groups][3 planes][64 rows][32 k], B [2 column groups][3][64][32] bf16 (rows of 64 bytes): the 4 x 2 waves of
// the workgroup share them as the GEMM's do; the 16-byte slots of a row are read XOR-swizzled by (row >> 2) & 3: conflict-free
// ds_read_b128 in both forms
__global__ __launch_bounds__(512) void probe32(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[6][3][64][16];   // A row groups 0..3, B column groups 4..5; plane; row; 32 bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ga = wave >> 1, gb = 4 + (wave & 1);
  for (int i = tid; i < 6 * 3 * 64 * 16; i += 512) ((uint32_t*)lds)[i] = seed[i];
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // two 16-deep steps per iteration
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = *(const bf16x8*)&lds[ga][p][i * 32 + fr][(((2 * ks + fh) ^ ((fr >> 2) & 3))) * 4];
          b[p][i] = *(const bf16x8*)&lds[gb][p][i * 32 + fr][(((2 * ks + fh) ^ ((fr >> 2) & 3))) * 4];
        }
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ __launch_bounds__(512) void probe16(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[6][3][64][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ga = wave >> 1, gb = 4 + (wave & 1);
  for (int i = tid; i < 6 * 3 * 64 * 16; i += 512) ((uint32_t*)lds)[i] = seed[i];
  __syncthreads();
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 15, fg = lane >> 4;   // row of the 16-row block, k group of 8
  for (int it = 0; it < iters; ++it) {
    bf16x8 a[3][4], b[3][4];   // one 32-deep step per iteration
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[p][i] = *(const bf16x8*)&lds[ga][p][i * 16 + fr][((fg ^ ((fr >> 2) & 3))) * 4];
        b[p][i] = *(const bf16x8*)&lds[gb][p][i * 16 + fr][((fg ^ ((fr >> 2) & 3))) * 4];
      }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
}

extern "C" int probe_run(int which, const void* seed, float* out, int blocks, int iters, void* stream) {
  if (which == 32) hipLaunchKernelGGL(probe32, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint32_t*)seed, out, iters);
  else hipLaunchKernelGGL(probe16, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint32_t*)seed, out, iters);
  return hipGetLastError() != hipSuccess;
}
