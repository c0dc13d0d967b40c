// fp32 GEMMs on the bf16 matrix pipe: every fp32 operand is carried as three bf16 pieces,
//
//   a = ah + am + al       ah = bf16(a), am = bf16(a - ah), al = bf16(a - ah - am)     (each subtraction is exact;
//                          3 x 8 significant bits hold the 24 of an fp32 mantissa),
//   a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + terms <= 2^-24 |a b|,
//
// six v_mfma_f32_32x32x16_bf16 per 16-deep k-step, every product exact, accumulated in fp32.
// The result is as close to the exact product as the fp32 MFMA's (measured against fp64 on the shapes below: rms error
// 3.4e-7 .. 1.4e-6 of the result's rms, fp32 MFMA 4.1e-7 .. 1.6e-6; scratch/bf16x3/, profiles/README.md "bf16x3") -
// this is fp32 arithmetic issued on a faster pipe, not a reduced precision: the bf16 pipe issues 16 x the MACs per clock
// of v_mfma_f32_32x32x2_f32, so six products cost 3/8 of the fp32 instruction's matrix cycles.
//
// Used for the 36 position GEMMs of Winograd F(4x4,3x3) (kernels_wino4.hip), D_g[t][n] = sum_c V_g[t][c] U_g[n][c]:
//
//   operands   three planes in K-CHUNK-MAJOR order, [3][G][K/16][R][16] bf16: the 32 rows x 32 bytes one LDS-DMA instruction
//              moves are 1 KB of consecutive memory.  V is written in this form by the input transform (wino4_in3), U once
//              per plan (split3).
//   kernel     256 x 128 output tile, twelve waves: eight compute (64 x 64 each = 2 x 2 MFMA tiles), four only issue the
//              buffer_load ... lds of the ring of four 36 KB stages (BK = 16), three stages ahead.  (A wave whose DMA
//              waits for a slot in the memory pipeline cannot issue MFMAs meanwhile; with loaders of their own the
//              K = 512 launches took 0.41 instead of 0.51 ms.)  The fragments of stage i + 1 are read while the 24 MFMAs
//              of stage i run, into the registers the products are done with (see `stage`); one barrier per stage.  The
//              kernel is persistent: whole rounds of tiles, left-over tiles cut in k (below).  The two 16-byte slots of a 32-byte row are
//              swapped in rows 8..15 of every 16, on the source side of the DMA and in the read: conflict-free
//              ds_read_b128 fragments.  Workgroup ids are mapped so that every XCD works through one contiguous eighth
//              of the (g, M tile, N tile) order: an operand tile crosses the fabric into one L2.
//   bound      the matrix pipe at the clock the chip sustains under bf16 MFMA load: a bare six-product loop issues 1.24-1.37
//              PFLOP/s of bf16 (2.5 nominal); in the plan the launches run at 0.85-1.25 = 140-210 TFLOP/s of fp32-equivalent
//              work against 120-136 of the fp32 MFMA kernel on the same GEMMs (1.37-1.63 x per launch).
#include "common.h"
#include "epilogue.h"

namespace kd {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = X3_BK;
constexpr int ROWB = BK * 2;                          // 32 bytes per LDS row
constexpr int NST = 4;
constexpr int STAGE_B = 3 * (BM + BN) * ROWB;         // 36 864 bytes
constexpr int B_OFF = 3 * BM * ROWB;

// m0 is not live across this statement (nothing else in the kernel uses it); the kernel orders its DMAs itself
__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// planes[p][g][k / 16][r][k % 16], p = 0 (high), 1 (middle), 2 (low), of x [G][R][K].  One thread: two consecutive k
// (v_cvt_pk_bf16_f32 rounds to nearest even)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, uint32_t* __restrict__ planes, int64_t n2,
                                                     int R, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const int K2 = K >> 1;
  const int k = (int)(i % K2) * 2;
  const int64_t gr = i / K2;
  const int r = (int)(gr % R);
  const int64_t g = gr / R;
  const int64_t o = (((g * (K / BK) + k / BK) * R + r) * BK + k % BK) >> 1;
  uint32_t h, m, l;
  x3_split(*(const f32x2*)(x + 2 * i), h, m, l);
  planes[o] = h;
  planes[n2 + o] = m;
  planes[2 * n2 + o] = l;
}

int launch_split3(const float* x, void* planes, int G, int R, int K, hipStream_t s) {
  KD_REQUIRE(K % BK == 0 && ((uintptr_t)x & 7) == 0, "bf16x3 planes need K % 16 == 0");
  const int64_t n2 = (int64_t)G * R * K / 2;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, x, (uint32_t*)planes, n2, R, K);
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// A3 [3][G][K/16][M][16], B3 [3][G][K/16][N][16] bf16; C [G][M][N] fp32.
//
// Persistent: P = min(tiles, CUs) workgroups.  Workgroup q (the q-th of its XCD's contiguous eighth, see below) computes
// tiles q, q + P, q + 2 P, ... of the (g, M tile, N tile) order - whole rounds, in which the workgroups that share an
// operand tile walk k in step and a k-chunk crosses the fabric into the XCD's L2 once.  (Contiguous stream-K runs were
// measured first and lost 40-60 % per stage: neighbouring workgroups sit 1/8 tile apart in k, the reuse distance of a
// chunk outgrows the 4 MB of L2.)  The R = tiles mod P tiles left over are cut in k into S = min(8, P / R, K / 128) parts,
// one per workgroup q < S R: without that the 288 tiles of the 16 x 16 level's GEMMs take two rounds on 256 CUs for 1.125
// rounds of work.  A part leaves its accumulators in a slab; `sum_slabs_kernel`, launched behind this one, adds a tile's
// slabs IN PART ORDER and stores the tile (the whole chip reads the 32 x 8 x 128 KB, 3-5 us).  (First form: the parts met
// inside the launch - slab, agent-scope release, a ticket from the tile's counter, the last arriver acquires and adds the
// slabs alone, nobody waits: correct under any interleaving of two launches, but one CU reading 1 MB at its L2 rate
// cost ~15 us at the end of every launch.)
// The LDS ring does not drain between tiles: the loaders run up to three stages into the next tile while the computing
// waves store the last one.
template <int V>
struct X3Set { static constexpr int value = V; };
struct X3Yes { static constexpr bool value = true; };
struct X3No { static constexpr bool value = false; };
constexpr int X3_MAX_WG = 512;       // slabs of the workspace
constexpr int X3_MAX_SPLIT = 8;

// A_F32: the A operand comes as plain fp32 [G][M][K] rows and the loader waves split it on its way into LDS (buffer_load to
// registers two stages ahead, nine VALU operations per pair of values, ds_write_b64 of the three planes) - the producer
// writes 4 instead of 6 bytes per element, and its writes are what bound it (kernels_wino4.hip)
// EPI: the token-GEMM / 1x1-conv form (G = 1): C rows with stride e.ldy, bias, residual, per-image gate (X3Epi, common.h);
// A rows (A_F32) with stride e.lda.  Without it the code is that of the Winograd position GEMMs, instruction for instruction.
// EK: 0 = none (the Winograd position GEMMs), 1 = bias / residual / gate / statistics with 16-byte accesses (token GEMMs,
// 1x1 skip convs: rows 16-byte aligned), 2 = the general form: 4-byte accesses, activation, PixelShuffle output (upsample convs) - kinds of their own so that the common one carries neither
// the extra branches nor the scalar registers of the rare one (the loop's scalar state spills to vector lanes: 73 spilled
// SGPRs with kind 1 alone, 133 when one kernel served both - and 791 instead of 645 us on the 256 -> 128 conv of the 256^2 map)
template <bool A_F32, int EK>
__global__ __launch_bounds__(768) void gemm_bf16x3_kernel(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int G, int M, int N, int K, int S_st,
                                                          float* __restrict__ slab, X3Epi e) {
  constexpr bool EPI = EK != 0;
  const int S = S_st & 0xff, stagger = S_st >> 8;
  const int lda = EPI ? e.lda : K, ldc = EPI ? e.ldy : N;
  // kind 1 turns its accumulator blocks through 2 KB of LDS per computing wave behind the ring (16-byte epilogue accesses):
  // 147456 + 16384 = 160 KB, the whole LDS of a CU
  constexpr int EPI_LDS = EK == 1 ? 8 * 2048 : 0;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE_B + EPI_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtiles = M / BM, ntiles = N / BN;
  const int nk = K / BK;
  const int P = gridDim.x;
  // XCD x (workgroup ids x, x + 8, ...) takes the x-th eighth of every round: neighbours in the tile order share an L2
  int q = blockIdx.x;
  if ((P & 7) == 0) q = (blockIdx.x & 7) * (P >> 3) + (blockIdx.x >> 3);
  const int tiles = mtiles * ntiles * G, rounds = tiles / P, R = tiles - rounds * P;
  // every other workgroup starts `stagger` x 4 us late (launches of many rounds whose tiles are short in k): the chip's
  // workgroups otherwise compute together and store together, the matrix pipes idle while HBM takes the tiles and HBM's
  // write side idle while they compute
  // (measured, same box, step of the headline plan: 26.21-26.23 ms without, 26.10 / 26.05-26.07 / 26.04-26.08 / 26.11-26.15
  // with 2 / 3 / 4 / 6 units; four phases (q & 3) 26.07-26.08)
  if (stagger && (q & 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const bool tail = q < S * R;                        // this workgroup has a part of a left-over tile
  const int part = tail ? q % S : 0;
  const int tail_k0 = part * nk / S, tail_k1 = (part + 1) * nk / S;
  const int nseg = rounds + (tail ? 1 : 0);
  const int T = rounds * nk + (tail ? tail_k1 - tail_k0 : 0);   // stage units of this workgroup
  auto seg_tile = [&](int sg) __attribute__((always_inline)) { return sg < rounds ? sg * P + q : rounds * P + q / S; };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  if (A_F32 && wave >= 8) {
    // loader l: rows 16 l .. 16 l + 15 of every 64-row block of the A tile (lane -> row = lane >> 2, quarter = lane & 3 of
    // its 64 bytes of a stage: 16 bytes, four values, per load; four loads per stage), and pieces 3 l .. 3 l + 2 of the 12
    // of the B planes (LDS-DMA, as in the other form).  Unit u is issued behind barrier u - 4 - its A values into registers
    // (set u % 4), its B pieces into ring buffer u % 4 - and must be in LDS before barrier u - 1: the A values are split and
    // written in front of it.  Three units in flight: 48 KB of A per CU.
    // The vmcnt waits are placed by hand and the A loads are inline assembly, because hipcc's own bookkeeping cannot count
    // across the `ic < T` branches: it made every use of a loaded value wait for ALL outstanding accesses - the round-4 / 5
    // form of this loader (builtin loads, "two units ahead") never had more than one stage in flight, whatever its source
    // said, and the launches that are short in k ran at 3.4-4 TB/s where the all-DMA form of the same GEMM moves 5.1-5.3.
    const int l = wave - 8;
    const uint32_t planeB = (uint32_t)((int64_t)G * N * K * 2);
    const i32x4 rsB = make_rsrc(B3, 3u * planeB);
    // (gather form: the resource spans the input map, 4 M pixels)
    const i32x4 rsA = make_rsrc(A3, (uint32_t)((int64_t)G * M * lda * 4 * ((EPI && e.a_tap_c > 0) ? 4 : 1)));
    const uint32_t voffB = (uint32_t)((lane >> 1) * ROWB + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
    const uint32_t chunkB = (uint32_t)(N * ROWB);
    const int arow = l * 16 + (lane >> 2), aq = lane & 3;
    // the lane's four rows of a tile (arow + 64 j): byte offsets of their first value.  Plain rows: fixed, the tile's base
    // goes into the scalar offset.  GATHER (X3Epi::a_tap_c = C > 0; kinds 1 / 2): the rows are the OUTPUT pixels of a 2 x 2 /
    // stride-2 conv (the Downsample's pixel-unshuffle + 1x1 conv) over a [B][Hi][Wi][lda] map and k = tap C + c
    // walks the four input pixels (tap = 2 dy + dx) of a row: the lane keeps the offset of pixel (2 oy, 2 ox) per row - set
    // per tile - and the tap's displacement (dy Wi + dx) lda is wave-uniform, so it joins the scalar offset.
    const bool gather = EPI && e.a_tap_c > 0;
    const int cpt = gather ? e.a_tap_c / BK : nk;   // stage units per tap
    constexpr int NR = 4, AD = 4, PER = NR + 3;     // accesses of a unit: NR loads, then 3 DMAs
    uint32_t voffAj[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) voffAj[j] = (uint32_t)(((arow + 64 * j) * lda + aq * 4) * 4);
    // LDS position of the lane's 8 bytes of a plane row: 16-byte slot (aq >> 1) ^ ((row >> 3) & 1), half aq & 1
    const int ldsA = arow * ROWB + ((((aq >> 1) ^ ((arow >> 3) & 1)) * 16) + (aq & 1) * 8);
    int ic = 0, iseg = 0, ik = 0, ik1 = 0, kc = 0, tap = 0;
    uint32_t baseA = 0, baseB = 0;
    auto locate = [&]() __attribute__((always_inline)) {
      const int it = seg_tile(iseg);
      const int nt = it % ntiles, mt = (it / ntiles) % mtiles, g = it / (ntiles * mtiles);
      baseA = (uint32_t)(((int64_t)g * M + mt * BM) * lda * 4);
      baseB = (uint32_t)((int64_t)g * N * K * 2) + (uint32_t)(nt * BN * ROWB);
      ik = iseg < rounds ? 0 : tail_k0;
      ik1 = iseg < rounds ? nk : tail_k1;
      if (gather) {
        const int Wo = e.a_wi >> 1, how = (e.a_hi >> 1) * Wo;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int m = mt * BM + arow + 64 * j;
          const int b = m / how, rem = m - b * how, oy = rem / Wo, ox = rem - oy * Wo;
          voffAj[j] = (uint32_t)((((b * e.a_hi + 2 * oy) * e.a_wi + 2 * ox) * lda + aq * 4) * 4);
        }
        baseA = 0;
        tap = ik / cpt;
        kc = ik - tap * cpt;
      }
    };
    locate();
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x4 ra[AD][NR];   // the A values of the units in flight
    auto issue_next = [&](auto SET) __attribute__((always_inline)) {
      constexpr int set = decltype(SET)::value;
      const int st = ic & (NST - 1);
      // plain: the tile's base + stage ik of the row; gather: the tap's pixel displacement + stage kc of the tap's channels
      const uint32_t soA = gather ? (uint32_t)((((tap >> 1) * e.a_wi + (tap & 1)) * lda + kc * BK) * 4) : baseA + (uint32_t)(ik * (BK * 4));
#pragma unroll
      for (int j = 0; j < NR; ++j)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(ra[set][j]) : "v"(voffAj[j]), "s"(rsA), "s"(soA) : "memory");
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int id = l * 3 + j, pl = id >> 2, pr = id & 3;
        const uint32_t dst = lds0 + (uint32_t)(st * STAGE_B + B_OFF + (pl * BN + pr * 32) * ROWB);
        dma16(rsB, dst, voffB, (uint32_t)pl * planeB + baseB + (uint32_t)ik * chunkB + (uint32_t)(pr * 32 * ROWB));
      }
      ++ic;
      if (++kc == cpt) {
        kc = 0;
        ++tap;
      }
      if (++ik == ik1 && ++iseg < nseg) locate();
    };
    // Unit u's A values have landed: accesses return in order; behind them its own 3 DMAs and the PER accesses of each of the
    // two later units may be in flight.  Short of two later units - the workgroup's last - wait for everything (a plain
    // statement).  ONE pinning statement on every path, the registers passing through it so that nothing that reads them
    // moves in front of it: with one statement per case hipcc joined the cases through COPIES of the registers, made in
    // front of the waits - of values still in flight.
    auto wait_a = [&](int u, u32x4(&r)[NR]) __attribute__((always_inline)) {
      if (ic < u + 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(3 + 2 * PER) : "memory");
    };
    auto wait_b = [&](int u) __attribute__((always_inline)) {   // ... and its B pieces
      if (ic < u + 3) wait_vm<0>(); else wait_vm<2 * PER>();
    };
    auto write_unit = [&](int u, auto SET) __attribute__((always_inline)) {   // split + store the A planes of unit u
      constexpr int set = decltype(SET)::value;
      wait_a(u, ra[set]);
      char* dst = lds + (u & (NST - 1)) * STAGE_B + ldsA;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        uint32_t h0, m0, l0, h1, m1, l1;
        const f32x2 v01 = {__uint_as_float(ra[set][j][0]), __uint_as_float(ra[set][j][1])};
        const f32x2 v23 = {__uint_as_float(ra[set][j][2]), __uint_as_float(ra[set][j][3])};
        x3_split(v01, h0, m0, l0);
        x3_split(v23, h1, m1, l1);
        *(u32x2*)(dst + j * 64 * ROWB) = u32x2{h0, h1};
        *(u32x2*)(dst + BM * ROWB + j * 64 * ROWB) = u32x2{m0, m1};
        *(u32x2*)(dst + 2 * BM * ROWB + j * 64 * ROWB) = u32x2{l0, l1};
      }
      wait_b(u);
    };
    if (ic < T) issue_next(X3Set<0>());
    if (ic < T) issue_next(X3Set<1>());
    if (ic < T) issue_next(X3Set<2>());
    if (ic < T) issue_next(X3Set<3>());
    write_unit(0, X3Set<0>());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // iteration c: unit c + 1 into LDS, barrier c, then unit c + 4 (into the registers and the ring buffer of unit c)
    auto iter = [&](int c, auto SET_W, auto SET_I) __attribute__((always_inline)) {   // sets of units c + 1 and c + 4
      if (c + 1 < T) write_unit(c + 1, SET_W);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (ic < T) issue_next(SET_I);
    };
    for (int c = 0; c < T; c += 4) {
      iter(c, X3Set<1>(), X3Set<0>());
      if (c + 1 < T) iter(c + 1, X3Set<2>(), X3Set<1>());
      if (c + 2 < T) iter(c + 2, X3Set<3>(), X3Set<2>());
      if (c + 3 < T) iter(c + 3, X3Set<0>(), X3Set<3>());
    }
    return;
  }
  if (wave >= 8) {
    // loader l moves pieces 9 l .. 9 l + 8 of the 36 of every stage (0..23: A plane id / 8, rows 32 (id % 8); 24..35: B
    // plane (id - 24) / 4, rows 32 ((id - 24) % 4)).  A piece = 32 rows x 32 bytes = 1 KB of consecutive memory: lane ->
    // LDS (row = lane >> 1, slot = lane & 1), read from source slot (lane & 1) ^ ((row >> 3) & 1) of the same row
    const int l = wave - 8;
    const uint32_t planeA = (uint32_t)((int64_t)G * M * K * 2), planeB = (uint32_t)((int64_t)G * N * K * 2);
    const i32x4 rsA = make_rsrc(A3, 3u * planeA), rsB = make_rsrc(B3, 3u * planeB);
    const uint32_t voff = (uint32_t)((lane >> 1) * ROWB + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
    const uint32_t chunkA = (uint32_t)(M * ROWB), chunkB = (uint32_t)(N * ROWB);
    // the next unit to fetch: number ic, stage ik of segment iseg (which ends before stage ik1)
    int ic = 0, iseg = 0, ik = 0, ik1 = 0;
    uint32_t baseA = 0, baseB = 0;   // byte offsets of the tile's rows of stage 0 in plane 0
    auto locate = [&]() __attribute__((always_inline)) {
      const int it = seg_tile(iseg);
      const int nt = it % ntiles, mt = (it / ntiles) % mtiles, g = it / (ntiles * mtiles);
      baseA = (uint32_t)((int64_t)g * M * K * 2) + (uint32_t)(mt * BM * ROWB);
      baseB = (uint32_t)((int64_t)g * N * K * 2) + (uint32_t)(nt * BN * ROWB);
      ik = iseg < rounds ? 0 : tail_k0;
      ik1 = iseg < rounds ? nk : tail_k1;
    };
    locate();
    auto issue_next = [&]() __attribute__((always_inline)) {
      const int st = ic & (NST - 1);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int id = l * 9 + j;
        const bool isA = id < 24;
        const int pl = isA ? id >> 3 : (id - 24) >> 2;
        const int pr = isA ? id & 7 : (id - 24) & 3;
        const uint32_t dst = lds0 + (uint32_t)(st * STAGE_B + (isA ? 0 : B_OFF) + (pl * (isA ? BM : BN) + pr * 32) * ROWB);
        const uint32_t soff = isA ? (uint32_t)pl * planeA + baseA + (uint32_t)ik * chunkA + (uint32_t)(pr * 32 * ROWB)
                                  : (uint32_t)pl * planeB + baseB + (uint32_t)ik * chunkB + (uint32_t)(pr * 32 * ROWB);
        dma16(isA ? rsA : rsB, dst, voff, soff);
      }
      ++ic;
      if (++ik == ik1 && ++iseg < nseg) locate();
    };
    // four units ahead: unit c + 4 goes into the buffer of unit c, whose fragments were all read during stage c - 1
    for (int j = 0; j < NST && ic < T; ++j) issue_next();
    if (T > 3) wait_vm<27>(); else if (T > 2) wait_vm<18>(); else if (T > 1) wait_vm<9>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < T; ++c) {   // the barriers of the computing waves' stages, one for one
#ifndef X3_EXP_NOWAIT   // (timing experiment of profiles/README.md: the loaders do not wait for their DMAs - wrong results)
      // unit c + 1 landed (units c + 2, c + 3 may be in flight)
      if (c + 3 < T) wait_vm<18>(); else if (c + 2 < T) wait_vm<9>(); else wait_vm<0>();
#endif
      __builtin_amdgcn_s_barrier();
      if (ic < T) issue_next();
    }
    return;
  }
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves of 64 x 64
  // (never zeroed: the first MFMAs of a segment take a zero C operand instead)
  f32x16 acc[2][2];
  const int fr = lane & 31, fh = lane >> 5;
  const int fslot = (fh ^ ((fr >> 3) & 1)) * 16;
  const char* fa = lds + (wm * 64 + fr) * ROWB + fslot;           // + stage, plane, 32-row block
  const char* fb = lds + B_OFF + (wn * 64 + fr) * ROWB + fslot;
  struct Frags {
    bf16x8 a[3][2], b[3][2];
  };
  auto read_a = [&](Frags& f, int c, int p) __attribute__((always_inline)) {
    const int so = (c & (NST - 1)) * STAGE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[p][i] = *(const bf16x8*)(fa + so + p * BM * ROWB + i * 32 * ROWB);
  };
  auto read_b = [&](Frags& f, int c, int p) __attribute__((always_inline)) {
    const int so = (c & (NST - 1)) * STAGE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) f.b[p][i] = *(const bf16x8*)(fb + so + p * BN * ROWB + i * 32 * ROWB);
  };
  // accumulators <-> slab: registers (i, j, 4 r4 .. 4 r4 + 3) of every lane side by side (1 KB rows per wave); slab q is
  // workgroup q's.  Buffer accesses, the lane's part in one VGPR and the rest in the scalar offset: no per-store address
  // registers (64-bit addresses of sixty-four stores get hoisted out of the stage loop and spilled)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsS =
      __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (int)((size_t)X3_MAX_WG * BM * BN * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
      (void*)C, 0, (int)((int64_t)G * M * ldc * 4 * ((EK == 2 && e.pixshuf_wo) ? 4 : 1)), 0x00020000);
  // the lane's part of the offsets, recomputed where it is used (from an opaque copy of the lane id: two registers less
  // held across the stage loop)
  auto lane_offsets = [&](int& vS, int& vC) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    vS = (wave * 64 + ln) * 16;
    vC = ((wm * 64 + 4 * (ln >> 5)) * ldc + wn * 64 + (ln & 31)) * 4;
  };
  auto put_slab = [&](int which) __attribute__((always_inline)) {
    int vS, vC;
    lane_offsets(vS, vC);
    int s0 = which * (BM * BN * 4);
    asm volatile("" : "+s"(s0));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const u32x4 v = {__float_as_uint(acc[i][j][4 * r4]), __float_as_uint(acc[i][j][4 * r4 + 1]),
                           __float_as_uint(acc[i][j][4 * r4 + 2]), __float_as_uint(acc[i][j][4 * r4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsS, vS, s0 + ((i * 2 + j) * 4 + r4) * (512 * 16), 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // C/D layout of 32x32 tiles: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  auto store_tile = [&](int tile) __attribute__((always_inline)) {
    int vS, vC;
    lane_offsets(vS, vC);
    const int nt = tile % ntiles, mt = (tile / ntiles) % mtiles, g = tile / (ntiles * mtiles);
    int Nv = ldc;
    asm volatile("" : "+s"(Nv));   // (opaque: the 64 scalar offsets below are not to be hoisted out of the loop)
    const int s0 = ((g * M + mt * BM) * Nv + nt * BN) * 4;   // (G M ldc floats < 2^29: gemm_bf16x3_ok)
    if constexpr (EK == 1) {
      // 16-byte epilogue (MI355X guide T21; epilogue.h does the same for conv_buf_kernel): every 32 x 32 accumulator block
      // goes through the wave's private 2 KB of LDS in two halves of 16 rows - 8 ds_write_b32, 2 ds_read_b128, no barrier -
      // and leaves as two 16-byte stores per half (+ two loads per added map): 16 vector-memory instructions per lane and
      // block instead of 64.  The short-K launches on the big maps (16 stages per tile) spent a third of a tile there.
      // Buffer accesses with the lane's part in one VGPR per map (no 64-bit address registers across the loop).
      typedef float f4 __attribute__((ext_vector_type(4)));
      int ln = lane;
      asm volatile("" : "+v"(ln));
      float* sc = (float*)(lds + NST * STAGE_B) + wave * 512;
      const int rr = ln >> 3, c4 = (ln & 7) * 4;              // read side: row rr (+ 8), columns c4 .. c4 + 3 of the half
      const int wcol = ln & 31, wrow = 4 * (ln >> 5);         // write side: column, first row of the lane's four
      const int colb = nt * BN + wn * 64, rowb0 = g * M + mt * BM + wm * 64;   // (g > 0: the Winograd position GEMMs, no added maps)
      const __amdgpu_buffer_rsrc_t rsR =
          __builtin_amdgcn_make_buffer_rsrc((void*)e.res, 0, e.res ? (int)((int64_t)M * e.ldres * 4) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsG =
          __builtin_amdgcn_make_buffer_rsrc((void*)e.gate_src, 0, e.gate_src ? (int)((int64_t)M * e.ldgs * 4) : 0, 0x00020000);
      const int vY = (rr * Nv + c4) * 4, vRw = (rr * e.ldres + c4) * 4, vGw = (rr * e.ldgs + c4) * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = colb + j * 32;   // first column of the block (the lane's: + c4)
        const f4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f4 b4 = e.bias ? *(const f4*)(e.bias + col + c4) : z4;
        const f4 g4 = e.gate ? *(const f4*)(e.gate + (int64_t)((mt * BM) / e.hw) * N + col + c4) : z4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float f1 = 0.f, f2 = 0.f;
          // the block's four rows of the added map(s), all issued here, in front of the block's LDS turns and stores: loaded
          // where they are used (round 5's first form) every one of the sixteen loads of a tile had `s_waitcnt vmcnt(0)` behind
          // it - which also waits for the STORES in front of it to be acknowledged: sixteen dependent round trips per tile,
          // 17 us of epilogue behind 20 us of MFMAs on the 256 -> 128 skip conv of the 256 x 256 map
          // (one register set: the plan never adds both maps to one launch - x3_linear_ok -; a caller of the C ABI that does
          // gets the residual loaded where it is used)
          const bool pre_g = e.gate_src != nullptr, pre_r = !pre_g && e.res != nullptr;
          u32x4 aq[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int row = rowb0 + i * 32 + 16 * h + 8 * t;   // (the lane's: + rr)
              if (pre_g) aq[h][t] = __builtin_amdgcn_raw_buffer_load_b128(rsG, vGw, (row * e.ldgs + col) * 4, 0);
              else if (pre_r) aq[h][t] = __builtin_amdgcn_raw_buffer_load_b128(rsR, vRw, (row * e.ldres + col) * 4, 0);
            }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
              for (int k = 0; k < 4; ++k) sc[(8 * q + wrow + k) * 32 + wcol] = acc[i][j][4 * (2 * h + q) + k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes have landed (nobody else reads them)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int row = rowb0 + i * 32 + 16 * h + 8 * t;   // (the lane's: + rr)
              f4 v = *(const f4*)(sc + (rr + 8 * t) * 32 + c4) + b4;
              const f4 a4 = {__uint_as_float(aq[h][t][0]), __uint_as_float(aq[h][t][1]), __uint_as_float(aq[h][t][2]),
                             __uint_as_float(aq[h][t][3])};
              if (pre_g) v += a4 * g4;
              if (pre_r) v += a4;
              if (pre_g && e.res) {
                const u32x4 rq = __builtin_amdgcn_raw_buffer_load_b128(rsR, vRw, (row * e.ldres + col) * 4, 0);
                v += f4{__uint_as_float(rq[0]), __uint_as_float(rq[1]), __uint_as_float(rq[2]), __uint_as_float(rq[3])};
              }
              const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
              __builtin_amdgcn_raw_buffer_store_b128(o, rsC, vY, (row * Nv + col) * 4, 0);
              f1 += (v[0] + v[1]) + (v[2] + v[3]);
              f2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], f2))));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next half's writes
          }
          if (e.seg) {
            // the lane's 16 values of 4 channels in fp32, fp64 from there on: lanes with equal (ln & 7) >> 2 hold the same
            // 16-channel segment - folded over the rows (lane bits 3-5) and the segment's four lanes (bits 0-1)
            double d1 = (double)f1, d2 = (double)f2;
#pragma unroll
            for (int off = 8; off <= 32; off <<= 1) {
              d1 += __shfl_xor(d1, off, 64);
              d2 += __shfl_xor(d2, off, 64);
            }
#pragma unroll
            for (int off = 1; off <= 2; off <<= 1) {
              d1 += __shfl_xor(d1, off, 64);
              d2 += __shfl_xor(d2, off, 64);
            }
            if ((ln & ~4) == 0) {   // lanes 0 and 4: columns 0-15 and 16-31 of the block
              const int rowb = rowb0 + i * 32, b = rowb / e.hw;
              const int seg = (col + 4 * (ln & 4) + e.seg_coff) >> 4;
              if (e.seg_rows8) {   // (a launch whose left-over tiles are cut in k: chunks of 8 rows, this block's sum in the first of its four)
                double* o = e.seg + (((int64_t)b * e.seg_nseg + seg) * (e.hw >> 3) + ((rowb - b * e.hw) >> 3)) * 2;
                o[0] = d1;
                o[1] = d2;
#pragma unroll
                for (int zz = 2; zz < 8; ++zz) o[zz] = 0.0;
              } else {
                double* o = e.seg + (((int64_t)b * e.seg_nseg + seg) * (e.hw >> 5) + ((rowb - b * e.hw) >> 5)) * 2;
                o[0] = d1;
                o[1] = d2;
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    if constexpr (EK == 2) {
      // y = acc + bias[col] (+ gate_src[row][col] gate[image][col]) (+ res[row][col]): a lane owns one column of each of its
      // two 32-column blocks, so bias and the gate are two scalars per lane; the added maps come in as 4-byte loads, four
      // per map in flight (the registers of the next tile's first fragments stay live across the epilogue)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int col0 = nt * BN + wn * 64 + (ln & 31);           // + 32 j
      const int row0 = mt * BM + wm * 64 + 4 * (ln >> 5);       // + 32 i + 8 r4 + k
      float bj[2] = {0.f, 0.f}, gj[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (e.bias) bj[j] = e.bias[col0 + j * 32];
        if (e.gate) gj[j] = e.gate[(int64_t)((mt * BM) / e.hw) * N + col0 + j * 32];   // (hw % 256 == 0: one image per tile)
      }
      const __amdgpu_buffer_rsrc_t rsR =
          __builtin_amdgcn_make_buffer_rsrc((void*)e.res, 0, e.res ? (int)((int64_t)M * e.ldres * 4) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsG =
          __builtin_amdgcn_make_buffer_rsrc((void*)e.gate_src, 0, e.gate_src ? (int)((int64_t)M * e.ldgs * 4) : 0, 0x00020000);
      const int vR = (row0 * e.ldres + col0) * 4, vG = (row0 * e.ldgs + col0) * 4;
      // PixelShuffle(2) form (upsample convs; weight rows packed n' = q Co + c, q = 2 i' + j'): the 32 columns of a block are
      // 32 channels of ONE sub-position q (Co % 32 == 0), its 32 rows 16-pixel runs of image rows (Wo % 16 == 0),
      // so the block lands on 32 output pixels two apart: out pixel (2 R + q / 2) 2 Wo + 2 ox + q % 2 of image row R
      const int Co = N >> 2, Wo = EK == 2 ? e.pixshuf_wo : 0;
      const int vP = (8 * (ln >> 5) * ldc + (ln & 31)) * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float f1 = 0.f, f2 = 0.f;
          const int rowb = mt * BM + wm * 64 + i * 32;
          // (the block's rows 0-15 and 16-31 are located separately: maps 16 pixels wide put them in two image rows)
          int q = 0, sPh[2] = {0, 0};
          if (EK == 2 && Wo) {
            const int cb = nt * BN + wn * 64 + j * 32;   // first column of the block
            q = cb / Co;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int rowh = rowb + 16 * hf, R = rowh / Wo, ox0 = rowh - R * Wo;
              sPh[hf] = (((2 * R + (q >> 1)) * (2 * Wo) + 2 * ox0 + (q & 1)) * ldc + (cb - q * Co)) * 4;
            }
          }
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float add[4] = {0.f, 0.f, 0.f, 0.f};
            if (e.res) {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                add[k] = __uint_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rsR, vR, ((i * 32 + 8 * r4 + k) * e.ldres + j * 32) * 4, 0));
            }
            if (e.gate_src) {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                add[k] = fmaf(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                  rsG, vG, ((i * 32 + 8 * r4 + k) * e.ldgs + j * 32) * 4, 0)),
                              gj[j], add[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float v = acc[i][j][4 * r4 + k] + bj[j];
              if (EK == 2 && e.act != ACT_NONE) v = ep_act(v, e.act);
              v += add[k];
              if (EK == 2 && Wo) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, vP, sPh[r4 >> 1] + 2 * (8 * (r4 & 1) + k) * Nv * 4, 0);
              else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, vC, s0 + ((i * 32 + 8 * r4 + k) * Nv + j * 32) * 4, 0);
              f1 += v;
              f2 = fmaf(v, v, f2);
            }
          }
          if (e.seg) {
            // GroupNorm partials of the stored 32 x 32 block for the layer that reads the map (SegSrc, common.h): the lane's
            // 16 values of one channel in fp32, fp64 from there on; the 16 lanes x 2 halves of a 16-channel segment are
            // folded with shuffles, lanes 0 / 16 write chunk (row of the image) / 32 - as conv_buf_kernel's epilogue does
            double d1 = (double)f1, d2 = (double)f2;
#pragma unroll
            for (int off = 1; off <= 8; off <<= 1) {
              d1 += __shfl_xor(d1, off, 64);
              d2 += __shfl_xor(d2, off, 64);
            }
            d1 += __shfl_xor(d1, 32, 64);
            d2 += __shfl_xor(d2, 32, 64);
            if ((ln & 47) == 0) {
              const int b = rowb / e.hw;
              if (EK == 2 && Wo) {   // four sub-positions per 32-row block: chunk 4 (block of the image) + q, channel c = column - q Co
                const int seg = (col0 + j * 32 - q * Co + e.seg_coff) >> 4;
                double* o = e.seg + (((int64_t)b * e.seg_nseg + seg) * ((e.hw >> 5) * 4) + ((rowb - b * e.hw) >> 5) * 4 + q) * 2;
                o[0] = d1;
                o[1] = d2;
              } else if (e.seg_rows8) {   // a launch whose left-over tiles are cut in k keeps a chunk per 8 rows (the summing
                                          // launch's granularity): this block's sum goes into the first of its four chunks
                const int seg = (col0 + j * 32 + e.seg_coff) >> 4;
                double* o = e.seg + (((int64_t)b * e.seg_nseg + seg) * (e.hw >> 3) + ((rowb - b * e.hw) >> 3)) * 2;
                o[0] = d1;
                o[1] = d2;
#pragma unroll
                for (int z = 2; z < 8; ++z) o[z] = 0.0;
              } else {
                const int seg = (col0 + j * 32 + e.seg_coff) >> 4;
                double* o = e.seg + (((int64_t)b * e.seg_nseg + seg) * (e.hw >> 5) + ((rowb - b * e.hw) >> 5)) * 2;
                o[0] = d1;
                o[1] = d2;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rsC, vC,
                                                s0 + ((i * 32 + (r & 3) + 8 * (r >> 2)) * Nv + j * 32) * 4, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  // one stage: barrier (unit c + 1 landed, the loaders may refill the buffer of unit c), the 24 MFMAs of unit c in
  // the order al bh, am bh, am bm, ah bh, ah bm, ah bl, and the fragments of unit c + 1:
  //   behind group 0:  al <- c + 1 (in place: the product is done with it);  ah, bh, bl <- c + 1 into second registers (the
  //                    last group still needs them; nothing is read behind group 5, the wait in front of group 0 would
  //                    expose it)
  //   behind group 2:  am <- c + 1 (in place)          behind group 4:  bm <- c + 1 (in place)
  //   behind group 5:  the second registers move into ah, bh, bl (24 register moves under the last MFMAs)
  // All of unit c + 1 is read during stage c, so the loaders may refill its buffer one stage later: a ring of four buffers,
  // the DMAs FOUR units ahead (with bl read one stage late - in place - they ran three ahead, and the computing waves spent
  // 13 % of a launch at the barrier waiting for DMAs to land)
  // so every operand read in place has at least three groups of MFMAs between its read and its next use, and the two
  // operands every group needs - the high pieces - a whole stage.  (A second set for all six operands, 48 registers more,
  // does not fit the 168 of a lane at three waves per SIMD once the loop carries the schedule's state; with all six read
  // in place - two groups of slack on ah and bh - a stage took 1.5 x the time.)
  // Reads past the last unit fetch a stale ring buffer nobody uses.
  // FRESH: the first unit of a segment - its first MFMAs take a zero C operand (a separate copy of the stage: a branch
  // inside it costs a conservative lgkmcnt(0) behind the reads that follow the first group)
  Frags f;
  bf16x8 ahn[2], bhn[2], bln[2];
  auto stage = [&](int c, auto FRESH) __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 0, 1, 0, 1, 2};
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[0]][i], f.b[PB[0]][j], decltype(FRESH)::value ? zero : acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(f, c + 1, 2);
    {
      const int so = ((c + 1) & (NST - 1)) * STAGE_B;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ahn[i] = *(const bf16x8*)(fa + so + i * 32 * ROWB);
        bhn[i] = *(const bf16x8*)(fb + so + i * 32 * ROWB);
        bln[i] = *(const bf16x8*)(fb + so + 2 * BN * ROWB + i * 32 * ROWB);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < 6; ++t) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA[t]][i], f.b[PB[t]][j], acc[i][j], 0, 0, 0);
      if (t == 2 || t == 4) {
        __builtin_amdgcn_sched_barrier(0);
        if (t == 2) read_a(f, c + 1, 1);
        if (t == 4) read_b(f, c + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.a[0][i] = ahn[i];
      f.b[0][i] = bhn[i];
      f.b[2][i] = bln[i];
    }
  };
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    read_a(f, 0, p);
    read_b(f, 0, p);
  }
  int c = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    const bool whole = sg < rounds;
    const int n = whole ? nk : tail_k1 - tail_k0, tile = seg_tile(sg);
    stage(c, X3Yes());
    for (int v = 1; v < n; ++v) stage(c + v, X3No());
    c += n;
    if (whole || S == 1) store_tile(tile);
    else put_slab(q);   // slab q = S x (left-over tile) + part
  }
}

bool gemm_bf16x3_ok(int G, int64_t M, int N, int K) {
  return G > 0 && M > 0 && M % BM == 0 && N % BN == 0 && K % 32 == 0 && (int64_t)3 * G * M * K * 2 < ((int64_t)1 << 32) &&
         (int64_t)G * M * K * 4 < ((int64_t)1 << 32) &&
         (int64_t)3 * G * N * K * 2 < ((int64_t)1 << 32) && (M / BM) * (int64_t)(N / BN) * G * (K / BK) < 0x7fffffff &&
         (int64_t)G * M * N < ((int64_t)1 << 29);
}

// The left-over tiles of gemm_bf16x3_kernel: tile t = slab[S t] + slab[S t + 1] + ... in that order.  A slab holds the
// accumulators of a 256 x 128 tile in register order: float4 index ((i 2 + j) 4 + r4) 512 + wave 64 + lane = rows
// wm 64 + i 32 + 8 r4 + 4 (lane >> 5) + 0..3, column wn 64 + j 32 + (lane & 31) (wm = wave >> 1, wn = wave & 1).
// Grid: 32 workgroups of 256 per left-over tile.  EPI: the epilogue of the token-GEMM form (X3Epi), as in the kernel.
template <bool EPI>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slab, float* __restrict__ C, int G, int M, int N,
                                                        int S, int first_tile, X3Epi e) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int t = blockIdx.x >> 5, el = (blockIdx.x & 31) * 256 + threadIdx.x;   // left-over tile, float4 of its slab
  const f32x4* s = (const f32x4*)(slab + (size_t)t * S * (BM * BN)) + el;
  // all parts in flight together, added in part order (as a run-time loop hipcc waited for every load before the next: S - 1
  // dependent round trips - most of the 10-16 us of these launches); parts past S re-read the last one and are not added
  f32x4 part[X3_MAX_SPLIT];
#pragma unroll
  for (int p = 0; p < X3_MAX_SPLIT; ++p) part[p] = s[(size_t)(p < S ? p : S - 1) * (BM * BN / 4)];
  f32x4 v = part[0];
#pragma unroll
  for (int p = 1; p < X3_MAX_SPLIT; ++p) v = p < S ? v + part[p] : v;
  const int lane = el & 63, wave = (el >> 6) & 7, r4 = (el >> 9) & 3, ij = el >> 11;
  const int mtiles = M / BM, ntiles = N / BN, tile = first_tile + t;
  const int nt = tile % ntiles, mt = (tile / ntiles) % mtiles, g = tile / (ntiles * mtiles);
  const int row = mt * BM + (wave >> 1) * 64 + (ij >> 1) * 32 + 8 * r4 + 4 * (lane >> 5);
  const int col = nt * BN + (wave & 1) * 64 + (ij & 1) * 32 + (lane & 31);
  const int ldc = EPI ? e.ldy : N;
  float* c = C + ((int64_t)g * M + row) * ldc + col;
  if constexpr (EPI) {
    const float b = e.bias ? e.bias[col] : 0.f;
    const float gt = e.gate ? e.gate[(int64_t)(row / e.hw) * N + col] : 0.f;
    float f1 = 0.f, f2 = 0.f;
    float radd[4] = {0.f, 0.f, 0.f, 0.f}, gsrc[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) radd[k] = e.res[(int64_t)(row + k) * e.ldres + col];
    }
    if (e.gate_src) {
#pragma unroll
      for (int k = 0; k < 4; ++k) gsrc[k] = e.gate_src[(int64_t)(row + k) * e.ldgs + col];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float add = radd[k];
      if (e.gate_src) add = fmaf(gsrc[k], gt, add);
      float o = v[k] + b;
      if (e.act != ACT_NONE) o = ep_act(o, e.act);
      o += add;
      c[(int64_t)k * ldc] = o;
      f1 += o;
      f2 = fmaf(o, o, f2);
    }
    if (e.seg) {
      // GroupNorm partials (SegSrc): the two lane halves hold rows 8 r4 .. + 3 and + 4 .. + 7 of the same columns, 16 lanes a
      // 16-channel segment - one chunk per EIGHT rows here (the 32 rows of an accumulator block sit in four workgroups)
      double d1 = (double)f1, d2 = (double)f2;
#pragma unroll
      for (int off = 1; off <= 8; off <<= 1) {
        d1 += __shfl_xor(d1, off, 64);
        d2 += __shfl_xor(d2, off, 64);
      }
      d1 += __shfl_xor(d1, 32, 64);
      d2 += __shfl_xor(d2, 32, 64);
      if ((lane & 47) == 0) {
        const int bi = row / e.hw;
        double* o = e.seg + (((int64_t)bi * e.seg_nseg + ((col + e.seg_coff) >> 4)) * (e.hw >> 3) + ((row - bi * e.hw) >> 3)) * 2;
        o[0] = d1;
        o[1] = d2;
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) c[(int64_t)k * N] = v[k];
}

static int x3_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return cus;
}

// Shape of a launch: P persistent workgroups; `rounds` whole rounds of P tiles; the R tiles left over (the first is tile
// `first`) are cut in k into S parts of an equal, even number of at least eight stages (S = 1: not cut), one part per
// workgroup q < S R.  With fewer tiles than CUs (token GEMMs of the 16 x 16 level: 64-128 tiles) there is no whole round:
// every tile is cut and P = S R <= CUs.
static void x3_shape(int G, int M, int N, int K, int* P, int* R, int* S, int* first) {
  const int tiles = (M / BM) * (N / BN) * G, nk = K / BK, cus = x3_cus();
  auto split_of = [&](int r, int p) {
    int s = p / r;
    if (s > X3_MAX_SPLIT) s = X3_MAX_SPLIT;
    if (s > nk / 8) s = nk / 8;
    while (s > 1 && (nk % (2 * s)) != 0) --s;
    return s < 1 ? 1 : s;
  };
  if (tiles < cus) {
    *R = tiles;
    *first = 0;
    *S = split_of(tiles, cus);
    *P = tiles * *S;
    return;
  }
  *P = cus;
  *R = tiles % cus;
  *first = tiles - *R;
  *S = *R ? split_of(*R, cus) : 1;
}

// workgroups of the launch
int gemm_bf16x3_workgroups(int G, int M, int N, int K) {
  int P, R, S, first;
  x3_shape(G, M, N, K, &P, &R, &S, &first);
  return P;
}
size_t gemm_bf16x3_workspace_bytes() {   // a slab per workgroup
  return (size_t)X3_MAX_WG * BM * BN * sizeof(float);
}

// rows per chunk of the GroupNorm partials a launch leaves (X3Epi::seg): 32 from the kernel's epilogue, 8 from the summing launch
int gemm_bf16x3_seg_rows(int M, int N, int K) { return gemm_bf16x3_needs_sum(1, M, N, K) ? 8 : 32; }
bool gemm_bf16x3_needs_sum(int G, int M, int N, int K) {
  int P, R, S, first;
  x3_shape(G, M, N, K, &P, &R, &S, &first);
  return R && S > 1;
}
int launch_gemm_bf16x3_sum(float* C, int G, int M, int N, int K, const void* ws, hipStream_t s, const X3Epi* epi) {
  int P, R, S, first;
  x3_shape(G, M, N, K, &P, &R, &S, &first);
  if (!(R && S > 1)) return 0;
  if (epi)
    hipLaunchKernelGGL(sum_slabs_kernel<true>, dim3((unsigned)(R * 32)), dim3(256), 0, s, (const float*)ws, C, G, M, N, S, first, *epi);
  else
    hipLaunchKernelGGL(sum_slabs_kernel<false>, dim3((unsigned)(R * 32)), dim3(256), 0, s, (const float*)ws, C, G, M, N, S, first,
                       X3Epi{});
  KD_HIP_CHECK(hipGetLastError());
  return 0;
}

// the epilogue form's extra conditions (G = 1): row strides that keep every byte offset below 2^31, one image per tile
// where a per-image gate is applied, 4-byte aligned maps
bool gemm_bf16x3_epi_ok(int64_t M, int N, int K, const X3Epi& e) {
  if (!gemm_bf16x3_ok(1, M, N, K)) return false;
  if ((e.a_tap_c ? e.lda < e.a_tap_c : e.lda < K) || (e.ldy < N && !e.pixshuf_wo) || (e.lda & 3)) return false;
  // gather form (2 x 2 / stride-2 conv): four taps of C channels, whole stage units per tap, even map sides, whole images
  if (e.a_tap_c && (K != 4 * e.a_tap_c || e.a_tap_c % BK || (e.a_wi & 1) || (e.a_hi & 1) || e.a_wi <= 0 || e.a_hi <= 0 ||
                    M % ((e.a_wi >> 1) * (int64_t)(e.a_hi >> 1)) || 4 * M * (int64_t)e.lda * 4 >= ((int64_t)1 << 32)))
    return false;
  if (M * (int64_t)e.lda * 4 >= ((int64_t)1 << 31) || M * (int64_t)e.ldy >= ((int64_t)1 << 29)) return false;
  if (e.res && (e.ldres < N || M * (int64_t)e.ldres * 4 >= ((int64_t)1 << 31))) return false;
  if (e.gate_src && (!e.gate || e.ldgs < N || e.hw <= 0 || e.hw % BM || M * (int64_t)e.ldgs * 4 >= ((int64_t)1 << 31))) return false;
  // output statistics: whole 32-row blocks per image, segments of 16 channels (a launch whose tiles are cut in k leaves
  // them from its summing launch, one chunk per 8 rows: gemm_bf16x3_seg_rows)
  if (e.seg && (e.hw <= 0 || e.hw % 32 || M % e.hw || (e.seg_coff & 15) || e.seg_nseg <= 0)) return false;
  // PixelShuffle(2) output: whole 16-pixel runs of an image row and 32-channel runs of a sub-position per accumulator block;
  // the kernel's epilogue only (no tile cut in k), no added maps
  if (e.pixshuf_wo && (e.pixshuf_wo % 16 || M % e.pixshuf_wo || (N & 127) || e.res || e.gate_src || gemm_bf16x3_needs_sum(1, (int)M, N, K) ||
                       4 * M * (int64_t)e.ldy >= ((int64_t)1 << 29) || e.ldy < N / 4))
    return false;
  if (e.act < ACT_NONE || e.act > ACT_SIGMOID) return false;
  return true;
}

// ws: gemm_bf16x3_workspace_bytes() bytes (the slabs of the left-over tiles' parts); one per stream of launches (a plan's
// launches are ordered on its stream)
// a_f32: A is plain fp32 [G][M][K] (split by the kernel's loader waves) instead of three planes
// with_sum = false: the caller launches launch_gemm_bf16x3_sum behind it (where gemm_bf16x3_needs_sum)
// epi != nullptr: the token-GEMM / 1x1-conv form (G = 1; gemm_bf16x3_epi_ok)
int launch_gemm_bf16x3(const void* A3, const void* B3, float* C, int G, int M, int N, int K, void* ws, hipStream_t s, bool a_f32,
                       bool with_sum, const X3Epi* epi) {
  KD_REQUIRE(gemm_bf16x3_ok(G, M, N, K), "bf16x3 GEMM needs M % 256 == 0, N % 128 == 0, K % 32 == 0 and operand planes < 4 GB");
  KD_REQUIRE((((uintptr_t)A3 | (uintptr_t)B3 | (uintptr_t)ws) & 15) == 0 && ws, "bf16x3 GEMM needs 16-byte aligned operand planes and a workspace");
  KD_REQUIRE(!epi || (G == 1 && gemm_bf16x3_epi_ok(M, N, K, *epi) && (a_f32 || (epi->lda == K && !epi->a_tap_c))),
             "bf16x3 GEMM, epilogue form: G = 1, row strides >= the row, byte offsets < 2^31, one image per 256-row tile under a gate");
  int P, R, S, first;
  x3_shape(G, M, N, K, &P, &R, &S, &first);
  KD_REQUIRE(P <= X3_MAX_WG, "bf16x3 GEMM: more workgroups than the workspace holds");
  float* slab = (float*)ws;
  X3Epi e = epi ? *epi : X3Epi{};
  e.seg_rows8 = R && S > 1;
  // 16-byte epilogue accesses: every row of y and of the added maps 16-byte aligned
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  // The plain form (Winograd position GEMMs, G > 1) stores its tiles through kind 1's 16-byte path too - nothing added, rows
  // of N floats: 16 store instructions per lane and tile instead of 64, what the K = 256 launches (16 stages per tile) gain
  // most from.  (`epi` stays null for the summing launch: sum_slabs_kernel<false>.)
  // (not with fp32 A - the Cout = 128 layers of the large maps: 26.60 -> 26.71-26.73 ms per step with it)
  const bool plain_wide = !epi && !a_f32 && al16(C) && kd_switch("KD_X3_WIDE_STORE", 1) != 0;
  if (plain_wide) {
    e.ldy = N;
    e.lda = K;
  }
  e.wide = (epi || plain_wide) && al16(C) && e.ldy % 4 == 0 && al16(e.bias) && (!e.res || (al16(e.res) && e.ldres % 4 == 0)) &&
           (!e.gate_src || (al16(e.gate_src) && al16(e.gate) && e.ldgs % 4 == 0));
  const dim3 grid((unsigned)P), block(768);
  const uint16_t *a = (const uint16_t*)A3, *b = (const uint16_t*)B3;
  const int kind = plain_wide && e.wide ? 1 : !epi ? 0 : (e.act != ACT_NONE || e.pixshuf_wo || !e.wide) ? 2 : 1;
  // every other workgroup starts 3 x 4 us late where a launch runs at least eight rounds of tiles that are short in k (see the
  // kernel): the position GEMMs and 1x1 convs of the 128 x 128 and 256 x 256 levels
  const int tiles_all = (M / BM) * (N / BN) * G;
  const int stag = tiles_all / P >= 8 && K <= 256 ? kd_switch("KD_X3_STAGGER", 3) : 0;
  const int S_st = S | (stag << 8);
#define KD_X3(AF, EKIND) hipLaunchKernelGGL((gemm_bf16x3_kernel<AF, EKIND>), grid, block, 0, s, a, b, C, G, M, N, K, S_st, slab, e)
  if (kind == 0) {
    if (a_f32) KD_X3(true, 0); else KD_X3(false, 0);
  } else if (kind == 1) {
    if (a_f32) KD_X3(true, 1); else KD_X3(false, 1);
  } else {
    if (a_f32) KD_X3(true, 2); else KD_X3(false, 2);
  }
#undef KD_X3
  KD_HIP_CHECK(hipGetLastError());
  if (with_sum && R && S > 1) return launch_gemm_bf16x3_sum(C, G, M, N, K, ws, s, epi);
  return 0;
}

}  // namespace kd
