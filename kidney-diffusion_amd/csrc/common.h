// Internal declarations shared by the engine's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

// Experiment switches of the A/B measurements logged in profiles/README.md.  The product build compiles every switch
// to its default (no behaviour depends on the ambient environment); `make EXTRA=-DKD_EXPERIMENT` builds a library
// whose switches read the environment (scratch/experiment_switch.h; kd_build_id() then ends in "+experiment").
#ifdef KD_EXPERIMENT
#include "../../scratch/experiment_switch.h"
#else
constexpr int kd_switch(const char*, int dflt) { return dflt; }
#endif

namespace kd {

void set_error(const std::string& msg);

#define KD_HIP_CHECK(expr)                                                               \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::kd::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + \
                      __FILE__ + ":" + std::to_string(__LINE__));                        \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define KD_REQUIRE(cond, msg)                                                             \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      ::kd::set_error(std::string("requirement failed: ") + #cond + " — " + (msg) + " at " + \
                      __FILE__ + ":" + std::to_string(__LINE__));                         \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_SIGMOID = 3 };
enum OutMode { OUT_NHWC = 0, OUT_PIXSHUF = 1, OUT_NCHW = 2 };

// Implicit-GEMM convolution / token GEMM.  y[m][n] = sum_{tap,c} x[pix(m,tap)][c] * w[tap][n][c]
struct ConvParams {
  const float* x;   // NHWC, channel stride ldx (>= Cin), 16-B aligned rows
  const float* w;   // packed [KH*KW][Cout][Cin]
  const float* bias;
  float* y;
  int B, Hi, Wi, Cin, ldx;
  int Ho, Wo, Cout;
  int KH, KW, stride, pad;
  int rr_cin;  // > 0: row-run mode, channels per input pixel (then KW == 1 and Cin == kernel_width * rr_cin)
  int act;
  const float* res;       // y += res[m][n]   (row stride ldres)
  int ldres;
  const float* gate_src;  // y += gate_src[m][n] * gate[b][n]   (resblock tail: h*gca + res_conv(x))
  const float* gate;      // [B][Cout]
  int ldgs;
  int out_mode;
  int ldy, yoff;          // OUT_NHWC: y[m*ldy + yoff + n]
  // batched 1x1 GEMM (Winograd positions): rows [z*wz_rows, (z+1)*wz_rows) use weight slab z of
  // wz_count slabs [Cout][Cin]; 0 = one weight tensor.  wz_rows must be a multiple of 128.
  int wz_rows, wz_count;
  // split-K for small-M layers: scratch [conv_ksplit(p)][M][Cout] floats, or nullptr (never split);
  // ksplit is filled in by launch_conv_igemm
  float* partial;
  int ksplit;
  // GroupNorm partials of the output for the layer that reads it (SegSrc): [B][seg_nseg][seg chunks][2] doubles with
  // segment (yoff + n - seg_c0) / 16; chunks: Ho*Wo/32 MFMA row tiles per image (x 4 sub-positions for OUT_PIXSHUF).
  // Needs Ho*Wo % 32 == 0, Cout % 16 == 0, yoff % 16 == 0, no split-K.  nullptr = off.
  double* seg_partial;
  int seg_nseg;
  int seg_c0;  // channel of y that is segment 0 of seg_partial (several launches can fill slices of one buffer)
  int wide_epilogue;  // set by launch_conv_igemm: 16-byte epilogue accesses (alignment checked there)
};

int launch_conv_igemm(const ConvParams& p, hipStream_t s);
// number of K splits launch_conv_igemm uses for this shape when `partial` is provided (1 = none)
int conv_ksplit(const ConvParams& p);
// chunks per image of the GroupNorm partials the launch can leave (ConvParams::seg_partial), 0 = it cannot
int conv_seg_chunks(const ConvParams& p);
int64_t conv_macs(const ConvParams& p);

// weight re-packing (device→device)
// OIHW -> [tap][O][Ipad] (zero for i >= I)
int launch_pack_oihw(const float* w_oihw, float* w_packed, int O, int I, int Ipad, int KH, int KW, hipStream_t s);
// OIHW -> row-run layout [KH][O][KW*Ipad] (zero for i >= I)
int launch_pack_oihw_rowrun(const float* w_oihw, float* w_packed, int O, int I, int Ipad, int KH, int KW, hipStream_t s);
// same, for the input-channel subset [a0,a0+na) U [b0,b0+nb) of an OIHW tensor with Itot input channels
int launch_pack_oihw_rowrun_sub(const float* w_oihw, float* w_packed, int O, int Itot, int a0, int na, int b0, int nb,
                                int Ipad, int KH, int KW, hipStream_t s);
// Downsample conv1x1 over pixel-unshuffled input ([O][4C], k = c*4+s1*2+s2) -> [tap=s1*2+s2][O][C]
int launch_pack_unshuffle(const float* w, float* w_packed, int O, int C, hipStream_t s);
// PixelShuffle conv1x1 ([4Co][I], n = c*4+i*2+j) -> rows n' = (i*2+j)*Co + c ; same for bias
int launch_pack_shuffle(const float* w, const float* b, float* w_packed, float* b_packed, int Co, int I,
                        hipStream_t s);

// ---- Winograd F(2x2,3x3) transforms (kernels_wino.hip)
// OIHW 3x3 weights -> U [16][O][I]
int launch_wino_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s);
// V[p][t][c] = (B^T d B)[p] of the 4x4 input tile of output tile t, d = SiLU(GroupNorm/FiLM(x)) when
// stats != nullptr (same arguments as launch_gn_apply_silu), d = x otherwise; zero padding outside.
// Both transforms work on a slice [t0, t0+nt) of the B*(H/2)*(W/2) output tiles; V and D are
// [16][nt][C] for that slice (KD_WINO_SLICE_MB bounds the workspace; one slice by default).
int launch_wino_in(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                   const float* scale_shift, int ld_ss, float* V, int B, int H, int W, int C, int G, int64_t t0,
                   int64_t nt, hipStream_t s);
// y[b][2ty+i][2tx+j][n] = (A^T D A)[i][j] + bias[n] (+ res)
// seg_partial != nullptr (C % 16 == 0): also the GroupNorm partials of y, layout [B][C/16][(H/2)(W/2)][2] doubles
int launch_wino_out(const float* D, const float* bias, const float* res, int ldres, float* y, int B, int H, int W,
                    int C, int64_t t0, int64_t nt, hipStream_t s, double* seg_partial = nullptr);

// ---- Winograd F(4x4,3x3) transforms (kernels_wino4.hip): U [36][O][I]; V [36][Mt][C] and D [36][Mt][N] over the
// Mt = B (H/4) (W/4) tiles of 4x4 outputs; arguments as launch_wino_in / launch_wino_out (one slice).  seg_partial != nullptr:
// wino4_out also leaves the GroupNorm partials of y, layout [B][N/16][(H/4)(W/4)][2] doubles (N % 64 == 0)
int launch_wino4_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s);
// mul_c0 >= 0 (GroupNorm form only): channels >= mul_c0 of x hold an unscaled skip tensor that the layer sees times `mul`
// (the statistics were taken of the scaled tensor): the factor goes into the folded affine, the map is not touched
int launch_wino4_in(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                    const float* scale_shift, int ld_ss, float* V, int B, int H, int W, int C, int G, hipStream_t s,
                    int mul_c0 = -1, float mul = 1.0f);
int launch_wino4_out(const float* D, const float* bias, const float* res, int ldres, float* y, int ldy, double* seg_partial,
                     int B, int H, int W, int C, hipStream_t s);

// ---- fp32 GEMMs on the bf16 matrix pipe (kernels_gemm_bf16x3.hip): operands as three bf16 planes in k-chunk-major order,
// planes[p][g][k / 16][row][k % 16], p = 0 high, 1 middle, 2 low piece (a = ah + am + al exactly); C[g][M][N] =
// A[g][M][K] B[g][N][K]^T with six bf16 MFMAs per k-step - fp32-class results (error against fp64 not above the fp32 MFMA's)
constexpr int X3_BK = 16;
// the three pieces of two consecutive values, each pair packed in a dword (v_cvt_pk_bf16_f32: round to nearest even;
// every subtraction is exact)
__device__ __forceinline__ void x3_split(const float __attribute__((ext_vector_type(2))) a, uint32_t& h, uint32_t& m,
                                         uint32_t& l) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 hb = __builtin_convertvector(a, b2);
  const f2 r1 = a - __builtin_convertvector(hb, f2);
  const b2 mb = __builtin_convertvector(r1, b2);
  const f2 r2 = r1 - __builtin_convertvector(mb, f2);
  const b2 lb = __builtin_convertvector(r2, b2);
  h = __builtin_bit_cast(uint32_t, hb);
  m = __builtin_bit_cast(uint32_t, mb);
  l = __builtin_bit_cast(uint32_t, lb);
}
bool gemm_bf16x3_ok(int G, int64_t M, int N, int K);   // M % 256 == 0, N % 128 == 0, K % 32 == 0, planes < 4 GB
int launch_split3(const float* x, void* planes, int G, int R, int K, hipStream_t s);   // x [G][R][K] -> 3 G R K bf16
// ws: gemm_bf16x3_workspace_bytes() bytes (the slabs of the left-over tiles' k-parts; no initial contents needed); one per
// stream of launches
size_t gemm_bf16x3_workspace_bytes();
// The token-GEMM / 1x1-conv form of the kernel (G = 1): y[m][n] = sum_k a[m][k] w[n][k] + bias[n] (+ gate_src[m][n]
// gate[m / hw][n]) (+ res[m][n]) with row strides; nullptr members are off
struct X3Epi {
  const float* bias = nullptr;
  const float* res = nullptr;
  const float* gate_src = nullptr;
  const float* gate = nullptr;   // [images][N]
  int ldres = 0, ldgs = 0;
  int hw = 0;                    // rows per image (gate), a multiple of 256
  int ldy = 0;                   // row stride of y (>= N)
  int lda = 0;                   // row stride of an fp32 A (a_f32); plane-form A is always dense
  // > 0 (a_f32 only): the 2 x 2 / stride-2 conv form - A is a [B][a_hi][a_wi][lda] map of a_tap_c channels, row m an OUTPUT
  // pixel, k = tap a_tap_c + c the channel c of its input pixel (2 oy + tap / 2, 2 ox + tap % 2); K = 4 a_tap_c
  int a_tap_c = 0, a_wi = 0, a_hi = 0;
  // GroupNorm partials of y for the layer that normalises it (SegSrc): [images][seg_nseg][hw / rows][2] doubles with rows =
  // gemm_bf16x3_seg_rows(M, N, K) (32, or 8 where the tiles are cut in k and the summing launch leaves them), segment
  // (n + seg_coff) / 16 of column n; needs hw % 32 == 0
  double* seg = nullptr;
  int seg_nseg = 0, seg_coff = 0;
  int seg_rows8 = 0;             // set by launch_gemm_bf16x3: chunks of 8 rows (gemm_bf16x3_seg_rows == 8)
  int wide = 0;                  // set by launch_gemm_bf16x3: 16-byte epilogue accesses (every row 16-byte aligned)
  int act = ACT_NONE;            // applied to acc + bias, before the added maps
  // > 0: PixelShuffle(2) output (upsample convs, weight rows packed n' = q N/4 + c): the width Wo of the INPUT map; y is
  // [4 M][ldy] rows of N / 4 channels; statistics chunks as conv_buf_kernel's: 4 (32-row block of the image) + q
  int pixshuf_wo = 0;
};
bool gemm_bf16x3_epi_ok(int64_t M, int N, int K, const X3Epi& e);
// a_f32: A is plain fp32 rows, split into its planes by the kernel's loader waves on the way into LDS
// The tiles left over after the whole rounds of the persistent kernel - all tiles when there are fewer than CUs - are cut
// in k; their parts' accumulators are added (and the epilogue applied) by a second launch (launch_gemm_bf16x3_sum;
// with_sum = true: launched here)
int launch_gemm_bf16x3(const void* A3, const void* B3, float* C, int G, int M, int N, int K, void* ws, hipStream_t s,
                       bool a_f32 = false, bool with_sum = true, const X3Epi* epi = nullptr);
bool gemm_bf16x3_needs_sum(int G, int M, int N, int K);
int gemm_bf16x3_seg_rows(int M, int N, int K);
int launch_gemm_bf16x3_sum(float* C, int G, int M, int N, int K, const void* ws, hipStream_t s, const X3Epi* epi = nullptr);
int gemm_bf16x3_workgroups(int G, int M, int N, int K);
// launch_wino4_in writing V as the three planes the bf16x3 GEMM reads ([3][36][C/16][Mt][16] bf16; Mt % 8 == 0, C % 16 == 0)
int launch_wino4_in3(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                     const float* scale_shift, int ld_ss, void* V3, int B, int H, int W, int C, int G, hipStream_t s,
                     int mul_c0 = -1, float mul = 1.0f);

// ---- fused Winograd F(2x2,3x3) conv + GroupNorm / FiLM / SiLU (kernels_wino_fused128.hip): items of 16 x 8 pixels x 128
// output channels, Cin <= 2048.  The kernel evaluates SiLU as u / (1 + 2^u) on u = -log2(e) (A x + B): the affine of
// launch_gn_fold / launch_gn_fold_seg carries WF_AB_SCALE and the weights packed for it WF_U_SCALE = -ln 2 (the conv is
// linear, so the factor left on the activated values moves into U)
constexpr float WF_AB_SCALE = -1.4426950408889634f;
constexpr float WF_U_SCALE = -0.6931471805599453f;
// ab = per-(image, channel) affine [B][C][2] of GroupNorm (+ FiLM): y = conv3x3(SiLU(A x + B)) + bias (+ res)
int launch_gn_fold(const float* stats, const float* gamma, const float* beta, const float* scale_shift, int ld_ss,
                   float* ab, int B, int C, int G, hipStream_t s);
bool wino_fused128_ok(int B, int H, int W, int C, int N);   // H % 8 == 0, W % 16 == 0, Cin % 4 == 0, Cin <= 2048, Cout % 128 == 0
int launch_wino_fused128_pack(const float* w_oihw, float* U, int O, int I, hipStream_t s, float scale);   // U: 16 N C floats
// items: id -> (image, y0, x0, slab), wino_fused128_items_count entries of 16 bytes for this (B, H, W, N)
size_t wino_fused128_items_count(int B, int H, int W, int N);
int launch_wino_fused128_items(void* items, int B, int H, int W, int N, hipStream_t s);
// out_partial != nullptr: the kernel also leaves (sum, sum of squares) partials of y per (image, group of out_groups)
// in launch_gn_finalize's layout, wino_fused_out_stats_chunks(H, W, N, G) entries per (image, group)
size_t wino_fused_out_stats_chunks(int H, int W, int N, int G);
int launch_wino_fused_gn128(const float* x, int ldx, const float* ab, const float* U, const float* bias, const float* res,
                            int ldres, float* y, int B, int H, int W, int C, int N, double* out_partial, int out_groups,
                            const void* items, hipStream_t s);
int wino_fused_gn_max_cin();   // largest Cin launch_wino_fused_gn128 takes (its affine table lives in LDS)
// stats[b][g] = (mean, rstd) from `chunks` (sum, sum of squares) partials per (b, g), summed in index order
int launch_gn_finalize(const double* partial, float* stats, int chunks, int B, int G, double count, float eps,
                       hipStream_t s);

// ---- norms / elementwise (kernels_norm.hip)
int launch_gn_stats(const float* x, int ldx, float* stats /*[B][G][2] mean,rstd*/, double* partial,
                    int B, int HW, int C, int G, float eps, hipStream_t s);
size_t gn_partial_bytes(int B, int HW, int C, int G);
int launch_gn_apply_silu(const float* x, int ldx, const float* stats, const float* gamma, const float* beta,
                         const float* scale_shift /*row b: [scale(C) | shift(C)], row stride ld_ss; or null*/,
                         int ld_ss, float* y, int B, int HW, int C, int G, hipStream_t s);
// y = LN(f(x))*g (+beta) (+res), f = in_act (ACT_GELU: the feed-forward's Linear -> GELU -> LayerNorm with the GEMM storing
// the raw product); g2 / y2 != nullptr: also y2 = LN(y)*g2 (the TransformerBlock's two LayerNorms around its residual)
int launch_layernorm(const float* x, int ldx, const float* g, const float* beta, const float* res, int ldres, float* y,
                     int rows, int C, float eps, hipStream_t s, int in_act = ACT_NONE, const float* g2 = nullptr,
                     float* y2 = nullptr, double* seg = nullptr, int seg_hw = 0, int planes = 0);
// (planes & 1 / & 2: y / y2 are written as the three bf16 planes of the bf16x3 GEMM's A operand, [3][C / 16][rows][16] bf16 -
// 6 rows C bytes - instead of fp32 rows: the GEMM that reads them needs no split by its loader waves)
// (seg != nullptr: also the GroupNorm partials of y, [rows / seg_hw][C / 16][seg_hw][2] doubles - one chunk per pixel)
// dst[row][0..C) = src[row][0..C) * scale, row strides ld_src / ld_dst (dst may be src)
int launch_copy_scale_rows(const float* src, int ld_src, float* dst, int ld_dst, int C, float scale, int64_t rows,
                           hipStream_t s);
// y[row] = [a[row] | b[row] * scale_b]; a == nullptr: only the b half is written (a's producer wrote in place)
int launch_concat2(const float* a, int Ca, const float* b, int Cb, float scale_b, float* y, int64_t rows,
                   hipStream_t s);
// y = a*gate[b][c] + r   (NHWC; a dense, r / y with row strides ldr / ldy); seg: optional segment partials of y,
// [B][C/16][gate_add_chunks(B, HW)][2] doubles (SegSrc below)
int launch_gate_add(const float* a, const float* gate, const float* r, int ldr, float* y, int ldy, double* seg, int B,
                    int HW, int C, hipStream_t s);
int gate_add_chunks(int B, int HW);

// GroupNorm statistics handed from the kernel that WRITES a map to the layer that normalises it: (sum, sum of
// squares) in fp64 per image, 16-channel segment and producer-defined chunk, [B][nseg][nchunk][2].  Producers:
// launch_gate_add, the conv epilogues (ConvParams::seg_partial), launch_wino_fused_gn128 (out_partial).
struct SegSrc {
  const double* partial;  // nullptr: unused
  int nseg, nchunk;
  int c0;                 // first channel of the normalised tensor this source covers
  float scale;            // the GroupNorm sees scale * x (skip connections: 2^-1/2)
  float ab_mul;           // factor folded into the affine's A for these channels (scale if the consumer reads x unscaled)
};
// stats [B][G][2] (mean, rstd) and / or ab [B][C][2] (launch_gn_fold's affine) from one or two sources
int launch_gn_fold_seg(SegSrc s0, SegSrc s1, const float* gamma, const float* beta, const float* scale_shift, int ld_ss,
                       float* ab, float* stats, int B, int C, int G, double count, float eps, hipStream_t s);
int launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s);
int launch_act(const float* a, float* y, int64_t n, int act, hipStream_t s);
// init image assembly: NCHW planes -> NHWC [B][H][W][Cpad] with channel order cond | x | lowres, zero pad
int launch_pack_init(const float* cond, int Cc, const float* x, const float* lowres, int Cl, float* y, int Cpad,
                     int B, int HW, hipStream_t s);
// y[b][r][:] rows copy helper for building key/value buffers: dst[b][row_off + r][c] = src[b][r][c]
int launch_copy_rows(const float* src, int64_t src_bstride, int ld_src, float* dst, int64_t dst_bstride,
                     int ld_dst, int rows, int C, int B, hipStream_t s);
// dst[b][row][c] = src[c] (broadcast one row to every batch)
int launch_bcast_row(const float* src, float* dst, int64_t dst_bstride, int C, int B, hipStream_t s);

// ---- attention / gca / skinny linear (kernels_attn.hip)
// q [B][Nq][H*D] with row stride ldq; out [B][Nq][H*D].  Keys = optional shared null key/value
// (64 floats each, same for every head and batch) followed by up to two segments
// k,v [B][n][Hkv*D] with row strides ld.
struct KVSeg {
  const float* k;
  const float* v;
  int ld;
  int n;
};
int launch_attention(const float* q, int ldq, const float* null_k, const float* null_v, KVSeg s0, KVSeg s1,
                     float* out, int ldo, int B, int Nq, int H, int Hkv, float scale, hipStream_t s);
// y[m][n] = act( sum_k f(x[m][k]) * w[n][k] + bias[n] ) for small M (<= 64); in_act applied to x
int launch_linear_gemv(const float* x, const float* w, const float* bias, float* y, int K, int N, int in_act, int act,
                       hipStream_t s);
int launch_linear_skinny(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int M,
                         int K, int N, int in_act, int act, hipStream_t s);
int launch_linear_skinny_valu(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int M,
                              int K, int N, int in_act, int act, hipStream_t s);
// GlobalContext: logits[b][p] = x[b][p][:]·wk + bk ; pooled[b][c] = sum_p softmax_p(logits) x[b][p][c]
int launch_gca_pool(const float* x, const float* wk, const float* bk, float* logits, float* pooled,
                    float* scratch, int B, int HW, int C, hipStream_t s);
size_t gca_scratch_floats(int B, int HW, int C);
// the whole gate of a GlobalContext block with C <= 512 in two launches (pooling partials; merge + FC + SiLU + FC + sigmoid)
bool gca_gate_fused_ok(int C, int hid);
int launch_gca_gate(const float* x, const float* wk, const float* bk, float* scratch, const float* w0, const float* b0, int hid,
                    const float* w2, const float* b2, float* gate, int B, int HW, int C, hipStream_t s);
// time embedding: out[b][0]=t, [1..h]=sin(t*w*2pi), [h+1..2h]=cos
int launch_sinu_emb(const float* t, const float* w, float* out, int B, int half, hipStream_t s);

// ---- the init cross-embed convs over x's 3 planes in one kernel (kernels_init.hip)
bool init_conv_fused_ok(int S, int n3, int n7, int n15);
size_t init_conv_weight_floats(int n3, int n7, int n15);
// w3 / w7 / w15: OIHW weights over Itot input channels, of which c0 .. c0 + 2 are x's planes
int launch_init_conv_pack(const float* w3, const float* w7, const float* w15, float* out, int n3, int n7, int n15, int Itot,
                          int c0, hipStream_t s);
// y[b][py][px][0 .. n3+n7+n15) (row stride ldy) = cat(conv3, conv7, conv15)(x) + (bias | res); seg: GroupNorm partials
// [B][C/16][S*S/32][2] or nullptr
int launch_init_conv(const float* x_nchw, const float* wp, const float* bias, const float* res, float* y, int ldy,
                     double* seg, int B, int S, int n3, int n7, int n15, hipStream_t s);

// ---- final conv to 3 channels (kernels_final.hip)
int launch_pack_final(const float* w_oihw, float* w_packed, int Ctot, int C, hipStream_t s);
int launch_final_static(const float* lowres, const float* w_oihw, const float* bias, float* stat, int Ctot, int c0,
                        int B, int H, int W, hipStream_t s);
int launch_final_gather(const float* P, const float* stat, const float* bias, float* out, int B, int H, int W,
                        hipStream_t s);

// ---- text conditioning helpers (kernels_text.hip)
int launch_text_select(const float* tok, const float* mask, const float* null_embed, float* out, int B, int L, int P,
                       int C, int drop, hipStream_t s);
int launch_add_rows_bcast(const float* x, const float* add, float* y, int B, int R, int C, hipStream_t s);
int launch_mean_rows(const float* x, float* y, int B, int R, int C, hipStream_t s);
// in place: each 64-float head segment of x[row][h*64 ..] -> x / max(||x||, 1e-12) (* scale_vec[64] if given)
int launch_l2norm_heads(float* x, int ld, int64_t rows, int heads, const float* scale_vec, hipStream_t s);
int launch_cfg_combine(const float* cond, const float* nul, float* out, float scale, int64_t n, hipStream_t s);

// ---- sampler (kernels_sampler.hip)
struct StepTables {  // device arrays of T floats
  const float *log_snr, *alpha, *sigma, *alpha_next, *sigma_next, *c, *noise_scale, *rn_a, *rn_b;
};
int launch_fill_time(const float* table, const int* d_iter, int R, float* out, int B, hipStream_t s);
int launch_x0(const float* x, const float* pred, float* x0, const StepTables& tb, const int* d_iter, int R,
              int objective, int64_t n, hipStream_t s);
int launch_quantile_abs(const float* x, float* out, int B, int64_t n, float q, void* ws, hipStream_t s);
size_t quantile_ws_bytes(int B);
int launch_ddpm_update(float* x, const float* x0, const float* s_thresh, const float* noise, int64_t noise_stride,
                       const uint64_t* d_seed, const StepTables& tb, const int* d_iter, int R, int dynamic_threshold,
                       int B, int64_t per, hipStream_t s);
int launch_inpaint_mix(float* x, const float* inp, const float* mask, const float* noise, int64_t noise_stride,
                       const uint64_t* d_seed, const StepTables& tb, const int* d_iter, int R, int B, int C, int64_t hw,
                       hipStream_t s);
// the Philox key lives in device memory (d_seed) so that one captured step graph serves every seed
int launch_seed_set(uint64_t* d_seed, uint64_t value, hipStream_t s);
int launch_renoise(float* x, const float* noise, int64_t noise_stride, const uint64_t* d_seed, const StepTables& tb,
                   const int* d_iter, int R, int T, int B, int64_t per, hipStream_t s);
int launch_iter_set(int* d_iter, int value, hipStream_t s);
int launch_cond_gather(const void* tab, void* dst, size_t bytes, const int* d_iter, int R, hipStream_t s);
// one tensor of the conditioning region, batch-major [B][row4 floats] at float offset off4; start = sum of row4 before it
struct CondSeg { uint32_t off4, row4, start; };
int launch_fill_time_rows(const float* table, int k0, int T, float* out, int B, hipStream_t s);
int launch_cond_scatter(const float* ws, float* tab, const CondSeg* d_segs, int nseg, uint32_t row_total, int B, int k0, int T,
                        int64_t cond_floats, hipStream_t s);
int launch_finalize(float* x, const float* inp, const float* mask, int B, int C, int64_t hw, hipStream_t s);
int launch_iter_inc(int* d_iter, hipStream_t s);
int launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, hipStream_t s);

}  // namespace kd
