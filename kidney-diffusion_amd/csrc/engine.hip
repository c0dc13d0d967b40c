// libkd_engine: UNet execution plan, sampler loop and the C ABI of include/kd_engine.h.
//
// The plan builder walks the same module tree the host-side `Unet` class declares
//  and emits a flat list of kernel launches with static shapes and
// static workspace offsets, so one denoising iteration can be captured into a hipGraph and
// replayed.  Feature maps are NHWC fp32; token tensors [B,N,C] are the same memory.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kd_engine.h"
#include "common.h"

namespace kd {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

#define KD_THROW_IF(expr)                                          \
  do {                                                             \
    if ((expr) != 0) throw std::runtime_error(::kd::g_err);        \
  } while (0)
#define KD_HIP_THROW(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));        \
  } while (0)

// ------------------------------------------------------------------------------ memory helpers
struct WeightPool {  // engine-owned HBM for weights, bump-allocated from 256 MiB slabs
  std::vector<void*> slabs;
  char* cur = nullptr;
  size_t left = 0, total = 0;
  float* alloc(size_t n_floats) {
    size_t bytes = (n_floats * sizeof(float) + 255) & ~size_t(255);
    if (bytes > left) {
      size_t slab = std::max(bytes, size_t(256) << 20);
      void* p = nullptr;
      KD_HIP_THROW(hipMalloc(&p, slab));
      slabs.push_back(p);
      cur = (char*)p;
      left = slab;
      total += slab;
    }
    float* r = (float*)cur;
    cur += bytes;
    left -= bytes;
    return r;
  }
  ~WeightPool() {
    for (void* p : slabs) (void)hipFree(p);
  }
};

// Packed weights of ONE UNet, shared by all its plans (batch / image-size variants made with
// kd_unet_create_shared): a packed form is produced the first time a plan asks for it.
struct WeightStore {
  WeightPool pool;
  std::unordered_map<std::string, float*> cache;
};

struct Arena {  // plan-time activation allocator with reuse (single in-order stream => safe)
  struct Blk {
    size_t off, size;
    bool free;
  };
  std::vector<Blk> blks;
  size_t end = 0, peak = 0;
  size_t alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (size_t i = 0; i < blks.size(); ++i) {
      if (blks[i].free && blks[i].size >= bytes) {
        if (blks[i].size > bytes) {
          Blk rest{blks[i].off + bytes, blks[i].size - bytes, true};
          blks[i].size = bytes;
          blks.insert(blks.begin() + i + 1, rest);
        }
        blks[i].free = false;
        return blks[i].off;
      }
    }
    if (!blks.empty() && blks.back().free) {  // grow the trailing free block
      blks.back().size = bytes;
      blks.back().free = false;
      end = blks.back().off + bytes;
      peak = std::max(peak, end);
      return blks.back().off;
    }
    blks.push_back(Blk{end, bytes, false});
    end += bytes;
    peak = std::max(peak, end);
    return blks.back().off;
  }
  void release(size_t off) {
    for (size_t i = 0; i < blks.size(); ++i) {
      if (blks[i].off == off && !blks[i].free) {
        blks[i].free = true;
        if (i + 1 < blks.size() && blks[i + 1].free) {
          blks[i].size += blks[i + 1].size;
          blks.erase(blks.begin() + i + 1);
        }
        if (i > 0 && blks[i - 1].free) {
          blks[i - 1].size += blks[i].size;
          blks.erase(blks.begin() + i);
        }
        return;
      }
    }
    throw std::runtime_error("arena: release of unknown offset");
  }
};

struct T {  // NHWC tensor in the workspace, or a channel slice of one (a skip tensor living in its concat buffer)
  size_t off = 0;  // arena block: what the reference counts key on
  int B = 0, H = 0, W = 0, C = 0;
  int ld = 0;      // row stride in floats; 0 = dense (C)
  int coff = 0;    // first channel inside the block's rows
  bool x3p = false;  // not fp32 rows: the three bf16 planes [3][C / 16][rows][16] a LayerNorm left for the bf16x3 GEMM that reads it
  int64_t rows() const { return (int64_t)B * H * W; }
  int HW() const { return H * W; }
  int LD() const { return ld ? ld : C; }
  size_t at() const { return off + (size_t)coff * sizeof(float); }  // byte offset of (row 0, channel 0)
  bool dense() const { return LD() == C; }
};

}  // namespace kd

using namespace kd;

// ------------------------------------------------------------------------------ the UNet object
struct kd_unet {
  kd_unet_config_t cfg;
  std::vector<std::function<int(hipStream_t)>> ops;
  std::vector<std::string> op_label;  // per-op description + algorithmic MACs (kd_unet_profile)
  std::vector<int64_t> op_macs;
  std::vector<int64_t> op_mfma;  // MACs the op issues on the matrix cores (Winograd: 16/36 of op_macs, padded K)
  std::shared_ptr<WeightStore> wstore;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  int64_t macs = 0;       // algorithmic MACs of one forward as the reference computes it
  int64_t mfma_macs = 0;  // MACs the per-step conv / GEMM launches issue on the matrix cores
  int64_t mfma_bf16_macs = 0;  // ... bf16 MACs of the bf16x3 GEMMs (six per fp32 MAC; not part of mfma_macs)
  int time_cond_dim = 0;
  // per-call I/O (read by the ops at run time)
  const float *in_x = nullptr, *in_lowres = nullptr, *in_cond = nullptr, *in_log_snr = nullptr,
              *in_lowres_log_snr = nullptr, *in_text_tokens = nullptr, *in_text_hiddens = nullptr;
  float* out = nullptr;
  // text-conditioning sub-plan (present iff cfg.cond_on_text && cfg.text_tokens > 0)
  std::vector<std::function<int(hipStream_t)>> text_ops;
  // step-invariant part of the forward (init conv over the cond / low-res planes): run once per
  // sampling call, before the captured per-step graph
  std::vector<std::function<int(hipStream_t)>> static_ops;
  int text_embed_dim = 0, max_text_len = 0;
  const float *in_text_embeds = nullptr, *in_text_mask = nullptr;
  int in_text_len = 0, in_text_drop = 0;
  float *out_text_tokens = nullptr, *out_text_hiddens = nullptr;
  float* s_pred_null = nullptr;  // classifier-free guidance: second forward's output
  // sampler scratch (allocated on first use)
  float *s_pred = nullptr, *s_x0 = nullptr, *s_thresh = nullptr, *s_time = nullptr, *s_tables = nullptr;
  int* s_iter = nullptr;
  uint64_t* s_seed = nullptr;  // Philox key, device-resident so the step graph does not depend on it
  void* s_qws = nullptr;
  int s_tables_cap = 0;
  std::vector<float> s_tables_host;  // what s_tables holds (9 x T): an unchanged schedule is not uploaded again
  float* s_tables_pinned = nullptr;  // staging for the asynchronous upload
  size_t s_tables_pinned_floats = 0;
  hipEvent_t s_tables_ev = nullptr;  // recorded behind the last upload from the staging buffer (whatever stream it ran on)
  // cached graph of one iteration
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t cap_stream = nullptr;
  std::vector<uint64_t> graph_key;

  // ---- step-invariant-per-schedule-index conditioning (time embeddings, FiLM scale / shift, time tokens and their
  // cross-attention K / V): ops flagged op_is_cond write only into the `cond_ws` region (offsets carry COND_FLAG); the
  // sampler can run them once per schedule step into `cond_tab` and replay a step with one gather instead
  static constexpr size_t COND_FLAG = size_t(1) << 62;
  std::vector<char> op_is_cond;
  char* cond_ws = nullptr;
  size_t cond_bytes = 0;
  char* cond_tab = nullptr;        // [T][cond_bytes]
  size_t cond_tab_bytes = 0;
  std::vector<float> cond_tab_sched;   // the schedule (9 x T) the table was built for
  int cond_tab_T = 0;
  float cond_tab_lowres = 0.f;
  bool cond_tab_valid = false;
  // the tensors of the cond region (every one batch-major: the table is then built B schedule steps per run)
  std::vector<kd::CondSeg> cond_segs;
  bool cond_rows_ok = true;
  kd::CondSeg* d_cond_segs = nullptr;
  uint32_t cond_row_total = 0;
  std::vector<char> cond_tab_row_ok;   // [T]: rows are built on demand, for the steps a call walks
  float cond_tab_build_ms = -1.f;    // device time and row count of the last build (kd_unet_cond_table_build_ms)
  int cond_tab_build_rows = 0, cond_tab_build_runs = 0;
  hipEvent_t cond_ev0 = nullptr, cond_ev1 = nullptr;   // around the last build; read lazily (cond_ev_pending)
  bool cond_ev_pending = false;
  int64_t cond_tab_refused_bytes = 0;   // > 0: the last sampling call wanted a table of this size and the cap / allocator refused
  void* x3_ws = nullptr;   // slabs of the bf16x3 GEMMs' left-over tiles (launch_gemm_bf16x3), allocated with the first such layer

  float* P(size_t off) const { return (float*)((off & COND_FLAG) ? cond_ws + (off & ~COND_FLAG) : ws + off); }
  ~kd_unet() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    void* frees[] = {ws, s_pred, s_x0, s_thresh, s_time, s_tables, s_iter, s_seed, s_qws, s_pred_null, cond_ws, cond_tab,
                     d_cond_segs, x3_ws};
    for (void* p : frees)
      if (p) (void)hipFree(p);
    if (s_tables_pinned) (void)hipHostFree(s_tables_pinned);
    if (s_tables_ev) (void)hipEventDestroy(s_tables_ev);
    if (cond_ev0) (void)hipEventDestroy(cond_ev0);
    if (cond_ev1) (void)hipEventDestroy(cond_ev1);
  }
};

namespace kd {

// ------------------------------------------------------------------------------ plan builder
struct Builder {
  kd_unet* u;
  Arena arena;
  std::unordered_map<std::string, std::pair<const float*, int64_t>> params;
  const kd_unet_config_t& cfg;
  int B;
  // batched time-MLP: every ResnetBlock's Linear(time_cond_dim, 2*dim_out) in one skinny GEMM
  std::vector<std::string> tmlp_prefixes;
  std::unordered_map<std::string, int> tmlp_off;
  int tmlp_total = 0;
  float *tmlp_w = nullptr, *tmlp_b = nullptr;
  T t_ss;  // [B, tmlp_total] scale|shift rows for all blocks
  // shared scratch
  T gn_stats_t, gn_partial_t;
  T init_x_;  // output of the init conv (set inside a nested scope of build())
  size_t gn_partial_max = 0;

  Builder(kd_unet* u_) : u(u_), cfg(u_->cfg), B(u_->cfg.batch) {}

  // ---- parameters
  bool has(const std::string& n) const { return params.count(n) != 0; }
  int64_t numel(const std::string& n) const {
    auto it = params.find(n);
    if (it == params.end()) throw std::runtime_error("missing parameter '" + n + "'");
    return it->second.second;
  }
  const float* raw(const std::string& n, int64_t expect = -1) {
    auto it = params.find(n);
    if (it == params.end()) throw std::runtime_error("missing parameter '" + n + "'");
    if (expect >= 0 && it->second.second != expect)
      throw std::runtime_error("parameter '" + n + "' has " + std::to_string(it->second.second) +
                               " elements, plan expects " + std::to_string(expect));
    return it->second.first;
  }
  // Engine-owned weight buffer `key` of this UNet: allocated and filled by `make(dst)` the first time any
  // plan of the UNet asks for it (the store is shared between plans, see kd_unet_create_shared).
  template <class F>
  float* cached(const std::string& key, size_t n_floats, F make) {
    auto& c = u->wstore->cache;
    auto it = c.find(key);
    if (it != c.end()) return it->second;
    float* dst = u->wstore->pool.alloc(n_floats);
    make(dst);
    c[key] = dst;
    return dst;
  }
  const float* P(const std::string& n, int64_t expect = -1) {  // engine-owned copy, torch layout
    const float* src = raw(n, expect);
    int64_t ne = numel(n);
    return cached("raw:" + n, (size_t)ne, [&](float* dst) {
      KD_HIP_THROW(hipMemcpyAsync(dst, src, (size_t)ne * sizeof(float), hipMemcpyDeviceToDevice, 0));
    });
  }
  const float* pack_conv(const std::string& n, int O, int I, int Ipad, int K) {
    const float* src = raw(n, (int64_t)O * I * K * K);
    return cached("tap:" + n + ":" + std::to_string(Ipad), (size_t)O * Ipad * K * K,
                  [&](float* dst) { KD_THROW_IF(launch_pack_oihw(src, dst, O, I, Ipad, K, K, 0)); });
  }

  const float* pack_conv_rowrun(const std::string& n, int O, int I, int Ipad, int K) {
    const float* src = raw(n, (int64_t)O * I * K * K);
    return cached("rowrun:" + n + ":" + std::to_string(Ipad), (size_t)O * Ipad * K * K,
                  [&](float* dst) { KD_THROW_IF(launch_pack_oihw_rowrun(src, dst, O, I, Ipad, K, K, 0)); });
  }

  const float* pack_conv_rowrun_sub(const std::string& n, int O, int Itot, int a0, int na, int b0, int nb, int Ipad,
                                    int K) {
    const float* src = raw(n, (int64_t)O * Itot * K * K);
    const std::string key = "rowrun_sub:" + n + ":" + std::to_string(a0) + "," + std::to_string(na) + "," +
                            std::to_string(b0) + "," + std::to_string(nb) + "," + std::to_string(Ipad);
    return cached(key, (size_t)O * Ipad * K * K, [&](float* dst) {
      KD_THROW_IF(launch_pack_oihw_rowrun_sub(src, dst, O, Itot, a0, na, b0, nb, Ipad, K, K, 0));
    });
  }

  // ---- activations
  // to_cond: the ops being emitted compute conditioning that depends on the inputs log_snr / lowres_log_snr / text only
  // (never on x): they are flagged, and everything they allocate lives in the permanent cond region (bump-allocated,
  // never reused), so the sampler can snapshot / restore that region per schedule step (kd_unet::cond_tab)
  bool to_cond = false;
  size_t cond_end = 0;
  size_t cond_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    const size_t off = cond_end;
    cond_end += bytes;
    return kd_unet::COND_FLAG | off;
  }
  T alloc(int b, int h, int w, int c) {
    T t;
    t.B = b; t.H = h; t.W = w; t.C = c;
    const size_t bytes = (size_t)b * h * w * c * sizeof(float);
    if (to_cond) {
      t.off = cond_alloc(bytes);
      // batch-major: [B][...] as allocated, or a linear()'s flattened [1][1][B x tokens][N] (the batch still outermost)
      const size_t total = (size_t)b * h * w * c, row = total / (size_t)B;
      if ((b == B || (b == 1 && h == 1 && w % B == 0)) && row > 0 && row * B == total && row < (size_t(1) << 31))
        u->cond_segs.push_back(CondSeg{(uint32_t)((t.off & ~kd_unet::COND_FLAG) / 4), (uint32_t)row, 0});
      else
        u->cond_rows_ok = false;
      return t;
    }
    t.off = arena.alloc(bytes);
    refs[t.off] = 1;
    return t;
  }
  T alloc_bytes(size_t bytes) {
    T t;
    t.B = 1; t.H = 1; t.W = 1; t.C = (int)((bytes + 3) / 4);
    if (to_cond) {
      t.off = cond_alloc(bytes);
      u->cond_rows_ok = false;   // a cond tensor without batch rows: the table is built step by step
      return t;
    }
    t.off = arena.alloc(bytes);
    refs[t.off] = 1;
    return t;
  }
  // every builder function returns a tensor owned by the caller (refcount 1) and never releases
  // its inputs; skip connections take an extra reference with retain().
  std::unordered_map<size_t, int> refs;
  void retain(const T& t) {
    if (t.off & kd_unet::COND_FLAG) return;
    refs[t.off] += 1;
  }
  void free(const T& t) {
    if (t.off & kd_unet::COND_FLAG) return;   // the cond region is permanent
    auto it = refs.find(t.off);
    if (it == refs.end() || it->second <= 0) throw std::runtime_error("plan: release of a dead tensor");
    if (--it->second == 0) {
      refs.erase(it);
      arena.release(t.off);
      drop_seg_block(t.off);
    }
  }
  // ---- GroupNorm partials handed from the kernel that writes a map to the layer that normalises it (SegSrc,
  // common.h): per tensor (keyed by its workspace offset) up to two channel ranges with their partial buffers.
  // KD_SEG_STATS=0 keeps the separate statistics passes (A/B, read per plan).
  struct SegPart {
    size_t off = 0;   // partial buffer [B][nseg][nchunk][2] doubles
    int nseg = 0, nchunk = 0, c0 = 0;
    float scale = 1.0f, ab_mul = 1.0f;
    bool owned = true;   // false: the buffer belongs to another tensor's list (a copied skip tensor) that outlives this one
  };
  struct SegList {
    size_t block = 0;   // arena block of the tensor: the partials die with it
    std::vector<SegPart> parts;
  };
  std::unordered_map<size_t, SegList> seg_of;   // key: T::at() - a slice and its buffer have different keys
  const bool seg_on = kd_switch("KD_SEG_STATS", 1) != 0;
  const bool x3_planes_on = kd_switch("KD_X3_PLANES", 1) != 0;   // LayerNorm outputs read by one bf16x3 GEMM: in plane form
  void drop_seg_block(size_t block) {
    for (auto it = seg_of.begin(); it != seg_of.end();) {
      if (it->second.block == block) {
        for (auto& sp : it->second.parts)
          if (sp.owned) arena.release(sp.off);
        it = seg_of.erase(it);
      } else {
        ++it;
      }
    }
  }
  // `to` is a copy of `from` that dies first: it may use from's partials
  void share_seg(const T& from, const T& to) {
    auto it = seg_of.find(from.at());
    if (it == seg_of.end()) return;
    SegList l = it->second;
    l.block = to.off;
    for (auto& sp : l.parts) sp.owned = false;
    seg_of[to.at()] = l;
  }
  // a skip slice joins the concat that holds it: its partials become channels [c0, ..) of the whole buffer `to`
  void move_seg(const T& from, const T& to, int c0, float scale, float ab_mul) {
    auto it = seg_of.find(from.at());
    if (it == seg_of.end()) return;
    SegList moved = it->second;
    seg_of.erase(it);
    SegList& dst = seg_of[to.at()];
    dst.block = to.off;
    for (auto sp : moved.parts) {
      sp.c0 += c0;
      sp.scale *= scale;
      sp.ab_mul *= ab_mul;
      dst.parts.push_back(sp);
    }
  }
  // reserves the partial buffer of channels [c0, c0 + 16 nseg) of tensor t; returns its workspace offset
  size_t add_seg(const T& t, int c0, int nseg, int nchunk, float scale = 1.0f, float ab_mul = 1.0f) {
    SegPart sp;
    sp.off = arena.alloc((size_t)t.B * nseg * nchunk * 2 * sizeof(double));
    sp.nseg = nseg; sp.nchunk = nchunk; sp.c0 = c0; sp.scale = scale; sp.ab_mul = ab_mul;
    SegList& l = seg_of[t.at()];
    l.block = t.off;
    l.parts.push_back(sp);
    return sp.off;
  }
  // the sources covering ALL C channels of x in order (at most two), or false
  bool seg_sources(const T& x, SegPart (&out)[2], int& n) const {
    n = 0;
    if (!seg_on || x.C % cfg.resnet_groups || (x.C / cfg.resnet_groups) % 16) return false;
    auto it = seg_of.find(x.at());
    if (it == seg_of.end()) return false;
    std::vector<SegPart> v = it->second.parts;
    std::sort(v.begin(), v.end(), [](const SegPart& a, const SegPart& b) { return a.c0 < b.c0; });
    int c = 0;
    for (auto& sp : v) {
      if (sp.c0 != c || n == 2) return false;
      out[n++] = sp;
      c += 16 * sp.nseg;
    }
    return c == x.C;
  }
  // emits the GroupNorm statistics of x into gn_stats_t - from the producer's partials when x has them (then
  // `ab` != nullptr also gets the folded affine in the same launch and true is returned), else by a pass over x
  bool emit_gn_stats(const T& x, const float* gamma, const float* beta, int ss_col, const T* ab) {
    const int G = cfg.resnet_groups, Bx = x.B, HW = x.HW(), C = x.C;
    SegPart sp[2];
    int n = 0;
    kd_unet* uu = u;
    const size_t so = gn_stats_t.off, sso = t_ss.off;
    const int ld = tmlp_total;
    if (seg_sources(x, sp, n)) {
      const SegPart a = sp[0], b2 = n > 1 ? sp[1] : SegPart();
      const bool two = n > 1, has_ab = ab != nullptr;
      const size_t abo = has_ab ? ab->off : 0;
      emit([=](hipStream_t s) {
        SegSrc s0{(const double*)uu->P(a.off), a.nseg, a.nchunk, a.c0, a.scale, a.ab_mul};
        SegSrc s1{two ? (const double*)uu->P(b2.off) : nullptr, b2.nseg, b2.nchunk, b2.c0, b2.scale, b2.ab_mul};
        const float* ssp = ss_col >= 0 ? uu->P(sso) + ss_col : nullptr;
        return launch_gn_fold_seg(s0, s1, gamma, beta, ssp, ld, has_ab ? uu->P(abo) : nullptr, uu->P(so), Bx, C, G,
                                  (double)HW * (C / G), 1e-5f, s);
      }, "gn fold seg HW" + std::to_string(HW) + " C" + std::to_string(C));
      return has_ab;
    }
    if (gn_partial_bytes(Bx, HW, C, G) > gn_partial_max) throw std::runtime_error("gn partial scratch too small");
    const size_t xo = x.at(), po = gn_partial_t.off;
    const int ldx = x.LD();
    emit([=](hipStream_t s) {
      return launch_gn_stats(uu->P(xo), ldx, uu->P(so), (double*)uu->P(po), Bx, HW, C, G, 1e-5f, s);
    }, "gn stats HW" + std::to_string(HW) + " C" + std::to_string(C));
    return false;
  }
  bool cond_hoist = true;  // (false inside the text sub-plan)
  bool to_text = false;    // building the text-conditioning sub-plan: ops go to u->text_ops
  bool to_static = false;  // emitting step-invariant work (run once per sampling call): u->static_ops
  const float* P_(const std::string& n) { return P(n); }
  void emit(std::function<int(hipStream_t)> f, std::string label = "op", int64_t macs = 0) {
    if (to_text) {
      u->text_ops.push_back(std::move(f));
      return;
    }
    if (to_static) {
      u->static_ops.push_back(std::move(f));
      return;
    }
    u->ops.push_back(std::move(f));
    u->op_label.push_back(std::move(label));
    u->op_macs.push_back(macs);
    u->op_mfma.push_back(0);
    u->op_is_cond.push_back(to_cond ? 1 : 0);
  }

  // ---- conv / GEMM emission
  struct ConvOpt {
    int act = ACT_NONE;
    const T* res = nullptr;       // y += res
    const T* gate_src = nullptr;  // y += gate_src * gate
    const T* gate = nullptr;
    int out_mode = OUT_NHWC;
    const T* dst = nullptr;  // write into channel slice [yoff, yoff+Cout) of dst
    int yoff = 0;
    int cin_logical = -1;
    bool out_external = false;  // OUT_NCHW into u->out
    bool rowrun = false;        // small-Cin wide-window conv: weights from pack_conv_rowrun
    int res_coff = 0;           // channel offset into res (res row stride stays res->C)
    int64_t macs_override = -1; // algorithmic MACs when the launch computes padded / re-associated work
    int wz_rows = 0;            // batched 1x1 GEMM over the Winograd positions (ConvParams::wz_rows)
    int wz_count = 16;          // ... 16 positions of F(2x2,3x3), 36 of F(4x4,3x3)
    bool want_seg = false;      // the output feeds a GroupNorm: leave its partials in the epilogue where possible
    int seg_c0 = -1, seg_cn = 0;  // channel range of y the partial buffer spans (default: this launch's own range);
                                  // launches filling slices of one tensor name the same span and share the buffer
  };
  // a 1x1 conv / token GEMM that runs on the bf16x3 kernel's epilogue form (kernels_gemm_bf16x3.hip)
  bool x3_linear_ok(const T& x, int Cout, int K, int stride, int pad, const ConvOpt& o) const {
    if (cfg.gemm_bf16x3 < 0 || cfg.conv_algo != 0 || cfg.x3_linear < 0 || to_text || to_static || to_cond) return false;
    if (K != 1 || stride != 1 || pad != 0 || o.rowrun || o.wz_rows || o.out_external) return false;
    if (o.out_mode != OUT_NHWC && o.out_mode != OUT_PIXSHUF) return false;
    if (o.gate_src && o.res) return false;
    // measured per launch against conv_buf_kernel at batch 16 (profiles/README.md, round 5): K >= 256 wins wherever the
    // launch runs whole rounds (256 -> 128 on the 256 x 256 map 748 -> 645 us); with fewer tiles than CUs every tile is cut
    // in k and a second launch adds the parts (10-15 us): K = 512 then only draws level (45.5 against 45.7 us), K >= 1024 wins
    // (K = 128 where the launch runs whole rounds: the 128 -> 512 upsample conv of the 128 x 128 map 455 -> 388 us)
    if (x.C % 32 || Cout % 128) return false;
    const int64_t M = x.rows();
    if (M % 256 || (M / 256) * (Cout / 128) < 64) return false;   // below 64 tiles the k-parts get too short
    const bool cut = gemm_bf16x3_needs_sum(1, (int)M, Cout, x.C);
    const int min_k = cfg.x3_linear > 0 ? cfg.x3_linear : cut ? 256 : 128;
    if (x.C < min_k) return false;
    if (cfg.x3_linear == 0 && x.C < 1024 && cut) return false;
    X3Epi e;
    e.lda = x.LD();
    if (x.x3p && (!x.dense() || x.coff)) return false;
    e.ldy = o.dst ? o.dst->LD() : (o.out_mode == OUT_PIXSHUF ? Cout / 4 : Cout);
    e.res = o.res ? (const float*)16 : nullptr;
    e.ldres = o.res ? o.res->LD() : 0;
    e.gate_src = o.gate_src ? (const float*)16 : nullptr;
    e.gate = o.gate_src ? (const float*)16 : nullptr;
    e.ldgs = o.gate_src ? o.gate_src->LD() : 0;
    e.hw = x.H * x.W;
    e.act = o.act;
    e.pixshuf_wo = o.out_mode == OUT_PIXSHUF ? x.W : 0;
    return gemm_bf16x3_epi_ok(M, Cout, x.C, e);
  }
  // images per launch of a 1x1 conv on the bf16x3 kernel: the whole batch, or - where the maps of the whole batch pass the
  // 2 GB its 32-bit offsets span (unet3's outer levels at batch 8) - the largest divisor of the batch that fits; 0 = not on it
  // (cfg.wino4_max_images, the test knob, cuts these launches too)
  int x3_linear_images(const T& x, int Cout, int K, int stride, int pad, const ConvOpt& o) const {
    const bool whole = x3_linear_ok(x, Cout, K, stride, pad, o);
    const int cap = cfg.wino4_max_images > 0 && whole ? cfg.wino4_max_images : x.B;
    if (whole && cap >= x.B) return x.B;
    if (x.B < 2 || x.x3p || cfg.x3_linear != 0) return whole ? x.B : 0;
    for (int ns = 2; ns <= x.B; ++ns) {
      if (x.B % ns) continue;
      T xs = x;
      xs.B = x.B / ns;
      if (xs.B <= cap && x3_linear_ok(xs, Cout, K, stride, pad, o)) return xs.B;
    }
    return whole ? x.B : 0;
  }
  T conv(const T& x, const float* w, const float* bias, int Cout, int K, int stride, int pad, const ConvOpt& o) {
    if (x.x3p && !x3_linear_ok(x, Cout, K, stride, pad, o))
      throw std::runtime_error("plan: a tensor in plane form reached a layer that is not a bf16x3 GEMM");
    int Ho = (x.H + 2 * pad - K) / stride + 1, Wo = (x.W + 2 * pad - K) / stride + 1;
    T y;
    if (o.dst) {
      y = *o.dst;
    } else if (o.out_mode == OUT_PIXSHUF) {
      y = alloc(x.B, 2 * Ho, 2 * Wo, Cout / 4);
    } else if (o.out_external) {
      y = T();
      y.B = x.B; y.H = Ho; y.W = Wo; y.C = Cout;
    } else {
      y = alloc(x.B, Ho, Wo, Cout);
    }
    ConvParams p{};
    p.w = w; p.bias = bias;
    p.B = x.B; p.Hi = x.H; p.Wi = x.W; p.Cin = x.C; p.ldx = x.LD();
    p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
    p.KH = K; p.KW = K; p.stride = stride; p.pad = pad;
    if (o.rowrun) {  // weights packed [KH][Cout][KW*C]: a kernel row is one contiguous K run (kernels_conv.hip)
      p.KW = 1;
      p.Cin = K * x.C;
      p.rr_cin = x.C;
    }
    p.wz_rows = o.wz_rows;
    p.wz_count = o.wz_rows > 0 ? o.wz_count : 0;
    p.act = o.act; p.out_mode = o.out_mode;
    p.ldy = (o.out_mode == OUT_NHWC || o.out_mode == OUT_PIXSHUF) ? y.LD() : 0;
    p.yoff = y.coff + o.yoff;   // (a slice destination: channels count from the slice's first)
    p.ldres = o.res ? o.res->LD() : 0;
    p.ldgs = o.gate_src ? o.gate_src->LD() : 0;
    size_t xo = x.at(), yo = y.off;
    bool has_res = o.res != nullptr, has_gs = o.gate_src != nullptr, ext = o.out_external;
    size_t ro = has_res ? o.res->at() : 0, gso = has_gs ? o.gate_src->at() : 0, go = has_gs ? o.gate->off : 0;
    kd_unet* uu = u;
    const int res_coff = o.res_coff, o_yoff = o.yoff;
    // small-M layers (batch-1 patches): split-K scratch, released right after the launch is recorded
    // (one in-order stream: the next op that reuses the block runs after the reduction)
    const int ks = conv_ksplit(p);
    T part;
    if (ks > 1) {   // (scratch of this launch pair, not conditioning state: never in the cond region / table)
      const bool tc = to_cond;
      to_cond = false;
      part = alloc_bytes((size_t)ks * x.B * Ho * Wo * Cout * sizeof(float));
      to_cond = tc;
    }
    const size_t parto = part.off;
    // GroupNorm partials of the output, left by the epilogue (channels [yoff, yoff + Cout) of y, or the Cout / 4
    // shuffled channels): several launches filling slices of one tensor (init conv) share the chunk count
    size_t sego = 0;
    int seg_nseg = 0, seg_c0 = 0, seg_nchunk = 0;
    if (o.want_seg && seg_on && !ext && !to_text && !to_static && !to_cond) {
      const int cw = o.out_mode == OUT_PIXSHUF ? Cout / 4 : Cout;
      seg_c0 = o.seg_c0 >= 0 ? o.seg_c0 : o.yoff;   // in channels of the tensor y (a slice counts from its own first)
      const int span = o.seg_c0 >= 0 ? o.seg_cn : cw;
      ConvParams probe = p;
      // (the residual's channel offset decides the alignment of its rows: checked here, at plan time)
      probe.res = has_res ? (const float*)(uintptr_t)(4096 + (o.res->at() % 4096) + (size_t)res_coff * sizeof(float)) : nullptr;
      probe.gate_src = has_gs ? (const float*)16 : nullptr;
      probe.seg_c0 = y.coff + seg_c0;
      probe.partial = ks > 1 ? (float*)16 : nullptr;   // split-K: statistics from the reduction kernel, one chunk per pixel
      // (the bf16x3 form of a 1x1 conv whose tiles are cut in k leaves its partials from the summing launch, a chunk per 8 rows)
      const int lin3_b = x3_linear_images(x, Cout, K, stride, pad, o);
      const bool lin3 = lin3_b > 0;
      int nchunk = conv_seg_chunks(probe) > 0 && lin3 && o.out_mode == OUT_NHWC
                       ? Ho * Wo / gemm_bf16x3_seg_rows(lin3_b * Ho * Wo, Cout, x.C)
                       : conv_seg_chunks(probe);
      // (the bf16x3 PixelShuffle epilogue takes maps 16 pixels wide - conv_buf_kernel's wants 32 - with the same chunks: four
      // sub-positions per 32 input pixels)
      if (lin3 && o.out_mode == OUT_PIXSHUF && nchunk == 0 && (Ho * Wo) % 32 == 0 && ((Cout >> 2) & 31) == 0)
        nchunk = (Ho * Wo / 32) * 4;
      if (nchunk > 0 && cw % 16 == 0 && span % 16 == 0 && o.yoff >= seg_c0 && o.yoff + cw <= seg_c0 + span) {
        SegPart* have = nullptr;
        auto it = seg_of.find(y.at());
        if (it != seg_of.end())
          for (auto& sp : it->second.parts)
            if (sp.c0 == seg_c0 && sp.nseg == span / 16 && sp.nchunk == nchunk) have = &sp;
        seg_nseg = span / 16;
        seg_nchunk = nchunk;
        sego = have ? have->off : add_seg(y, seg_c0, seg_nseg, nchunk);
      }
    }
    // ---- token GEMMs / 1x1 convs with K >= 512 as fp32 products on the bf16 matrix pipe (kernels_gemm_bf16x3.hip,
    // epilogue form: bias / residual / gate, strided rows; weights split into planes once per plan, the fp32 activations by
    // the kernel's loader waves; GroupNorm partials of the output where no tile is cut in k).  Layers with an activation
    // stay on conv_buf_kernel; cfg.conv_algo != 0 (the direct-convolution plans of the tests) too
    if (const int Bset = x3_linear_images(x, Cout, K, stride, pad, o)) {
      const int64_t M = (int64_t)Bset * Ho * Wo;   // rows of one launch: Bset images (the whole batch but for maps past 2 GB)
      const int Cin = x.C;
      const float* W3 = cached("x3lin:" + std::to_string((uintptr_t)w) + ":" + std::to_string(Cout) + "x" + std::to_string(Cin),
                               ((size_t)Cout * Cin * 3 + 1) / 2,
                               [&](float* dst) { KD_THROW_IF(launch_split3(w, dst, 1, Cout, Cin, 0)); });
      if (!u->x3_ws) KD_HIP_THROW(hipMalloc(&u->x3_ws, gemm_bf16x3_workspace_bytes()));
      X3Epi base;
      base.bias = bias;
      base.ldres = p.ldres;
      base.ldgs = p.ldgs;
      base.hw = Ho * Wo;
      base.ldy = p.ldy;
      base.lda = p.ldx;
      base.act = o.act;
      base.pixshuf_wo = o.out_mode == OUT_PIXSHUF ? Wo : 0;
      const int yoff = p.yoff;
      const int seg_coff = o_yoff - seg_c0;
      const int64_t seg_per_image = (int64_t)seg_nseg * seg_nchunk * 2;   // doubles: [image][segment][chunk][2]
      const int64_t y_per_image = (int64_t)(o.out_mode == OUT_PIXSHUF ? 4 : 1) * Ho * Wo * p.ldy;   // floats
      for (int b0 = 0; b0 < x.B; b0 += Bset) {
        // (every per-image pointer moved on by the set's first image)
        const size_t xo_s = xo + (size_t)b0 * x.H * x.W * p.ldx * sizeof(float), yo_s = yo + (size_t)b0 * y_per_image * sizeof(float);
        const size_t ro_s = ro + (size_t)b0 * Ho * Wo * p.ldres * sizeof(float), gso_s = gso + (size_t)b0 * Ho * Wo * p.ldgs * sizeof(float);
        const size_t go_s = go + (size_t)b0 * Cout * sizeof(float), sego_s = sego + (size_t)b0 * seg_per_image * sizeof(double);
        auto epi_of = [=]() {
          X3Epi e = base;
          e.res = has_res ? uu->P(ro_s) + res_coff : nullptr;
          e.gate_src = has_gs ? uu->P(gso_s) : nullptr;
          e.gate = has_gs ? uu->P(go_s) : nullptr;
          if (seg_nseg) {   // the output feeds a GroupNorm: its partials from the epilogue (of the summing launch where tiles are cut in k)
            e.seg = (double*)uu->P(sego_s);
            e.seg_nseg = seg_nseg;
            e.seg_coff = seg_coff;
          }
          return e;
        };
        const std::string shape = " M" + std::to_string(M) + " Cin" + std::to_string(Cin) + " Cout" + std::to_string(Cout);
        const int64_t m = o.macs_override >= 0 ? o.macs_override : M * Cout * Cin;
        const bool a_f32 = !x.x3p;   // (planes: written by the LayerNorm in front, the loader waves only move them)
        emit([=](hipStream_t s) {
          const X3Epi e = epi_of();
          return launch_gemm_bf16x3(uu->P(xo_s), W3, uu->P(yo_s) + yoff, 1, (int)M, Cout, Cin, uu->x3_ws, s, a_f32, false, &e);
        }, "conv k1 x3" + shape, m);
        u->macs += m;
        u->op_mfma.back() = 6 * M * Cout * Cin;   // bf16 MACs
        u->mfma_bf16_macs += u->op_mfma.back();
        if (gemm_bf16x3_needs_sum(1, (int)M, Cout, Cin))   // every tile cut in k (fewer tiles than CUs): the parts are added, and the epilogue applied, here
          emit([=](hipStream_t s) {
            const X3Epi e = epi_of();
            return launch_gemm_bf16x3_sum(uu->P(yo_s) + yoff, 1, (int)M, Cout, Cin, uu->x3_ws, s, &e);
          }, "conv k1 x3 sum" + shape);
      }   // sets of images
      if (ks > 1) free(part);
      return y;
    }
    emit([=](hipStream_t s) {
      ConvParams q = p;
      q.x = uu->P(xo);
      q.y = ext ? uu->out : uu->P(yo);
      q.res = has_res ? uu->P(ro) + res_coff : nullptr;
      q.gate_src = has_gs ? uu->P(gso) : nullptr;
      q.gate = has_gs ? uu->P(go) : nullptr;
      q.partial = ks > 1 ? uu->P(parto) : nullptr;
      if (seg_nseg) {
        q.seg_partial = (double*)uu->P(sego);
        q.seg_nseg = seg_nseg;
        q.seg_c0 = p.yoff - o_yoff + seg_c0;   // block channel of the partial buffer's segment 0
      }
      return launch_conv_igemm(q, s);
    });
    if (ks > 1) free(part);
    int cin = o.cin_logical > 0 ? o.cin_logical : x.C;
    int64_t m = o.macs_override >= 0 ? o.macs_override : (int64_t)x.B * Ho * Wo * Cout * cin * K * K;
    if (!to_text && !to_static) {
      u->op_label.back() = "conv k" + std::to_string(K) + " s" + std::to_string(stride) + " M" +
                           std::to_string((int64_t)x.B * Ho * Wo) + " Cin" + std::to_string(x.C) + " Cout" +
                           std::to_string(Cout);
      u->op_macs.back() = m;
    }
    // algorithmic MACs of one forward as the reference computes it: the step-invariant part of the
    // init conv is counted even though the engine runs it once per sampling call instead of per step
    if (!to_text) u->macs += m;
    if (!to_text && !to_static) {
      const int64_t issued = (int64_t)x.B * Ho * Wo * Cout * p.Cin * p.KH * p.KW;
      u->mfma_macs += issued;
      u->op_mfma.back() = issued;
    }
    return y;
  }
  // token GEMM y[M,N] = x[M,K] @ w[N,K]^T
  // `dst`: write into this tensor (a slice of a wider buffer: a skip tensor in its concat slot) instead of a new one
  T linear(const T& x, const float* w, const float* bias, int N, int act = ACT_NONE, const T* res = nullptr,
           const T* dst = nullptr) {
    T xf = x;
    xf.B = 1; xf.H = 1; xf.W = (int)x.rows();
    ConvOpt o;
    o.act = act;
    T rf, df;
    if (res) {
      rf = *res;
      rf.B = 1; rf.H = 1; rf.W = (int)res->rows();
      o.res = &rf;
    }
    if (dst) {
      df = *dst;
      df.B = 1; df.H = 1; df.W = (int)dst->rows();
      o.dst = &df;
    }
    T y = conv(xf, w, bias, N, 1, 1, 0, o);
    y.B = x.B; y.H = x.H; y.W = x.W;
    return y;
  }
  // small-M linear on [M,K] rows living at workspace offset (row stride ldx) -> [M,N] (row stride ldy)
  void skinny(size_t x_off, int ldx, const float* w, const float* bias, size_t y_off, int ldy, int M, int K, int N,
              int in_act, int act) {
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      return launch_linear_skinny(uu->P(x_off), ldx, w, bias, uu->P(y_off), ldy, M, K, N, in_act, act, s);
    }, "skinny M" + std::to_string(M) + " K" + std::to_string(K) + " N" + std::to_string(N), (int64_t)M * K * N);
    if (!to_text) u->macs += (int64_t)M * K * N;
  }

  // in_act: applied to x on the way in (ACT_GELU: the feed-forward's GELU when its GEMM stores the raw product);
  // g2 / y2: a second LayerNorm of the result in the same pass (y2 = LN(y) g2)
  // planes & 1 / & 2: y / y2 is read by one bf16x3 GEMM and nothing else - left as that kernel's three bf16 planes
  T layernorm(const T& x, const float* g, const float* beta, const T* res = nullptr, int in_act = ACT_NONE,
              const float* g2 = nullptr, T* y2 = nullptr, bool want_seg = false, int planes = 0) {
    if (!x3_planes_on || x.C % 16 || x.C > 4096 || to_cond || to_text || to_static) planes = 0;
    auto alloc_out = [&](bool pl) {
      if (!pl) return alloc(x.B, x.H, x.W, x.C);
      T t = alloc_bytes((size_t)x.rows() * x.C * 6);
      t.B = x.B; t.H = x.H; t.W = x.W; t.C = x.C;
      t.x3p = true;
      return t;
    };
    if ((planes & 1) && want_seg) throw std::runtime_error("plan: GroupNorm partials of a LayerNorm output in plane form");
    T y = alloc_out(planes & 1);
    if (g2) *y2 = alloc_out(planes & 2);
    size_t xo = x.at(), yo = y.off, ro = res ? res->at() : 0, y2o = g2 ? y2->off : 0;
    bool hr = res != nullptr;
    int rows = (int)x.rows(), C = x.C, ldx = x.LD(), ldres = res ? res->LD() : 0;
    // the output feeds a GroupNorm (the ResnetBlock's block2 behind its cross-attention): one chunk of partials per pixel
    const int hw = x.HW();
    const bool sg = want_seg && seg_on && !to_cond && !to_text && !to_static && C % 16 == 0 && C <= 4096 && x.B * hw == rows;
    const size_t sgo = sg ? add_seg(y, 0, C / 16, hw) : 0;
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      return launch_layernorm(uu->P(xo), ldx, g, beta, hr ? uu->P(ro) : nullptr, ldres, uu->P(yo), rows, C, 1e-5f, s, in_act, g2,
                              g2 ? uu->P(y2o) : nullptr, sg ? (double*)uu->P(sgo) : nullptr, hw, planes);
    }, std::string(g2 ? "ln x2 rows" : "ln rows") + std::to_string(rows) + " C" + std::to_string(C) + (planes ? " planes" : ""));
    return y;
  }
  // would linear(x, .., N) run on the bf16x3 kernel (x3_linear_ok on the flattened rows)?
  // (x stands for a dense tensor of its shape: the LayerNorm output the GEMM will read)
  bool linear_is_x3(const T& x, int N, const T* res = nullptr, const T* dst = nullptr) const {
    T xf = x;
    xf.B = 1; xf.H = 1; xf.W = (int)x.rows();
    xf.ld = 0; xf.coff = 0;
    ConvOpt o;
    T rf, df;
    if (res) {
      rf = *res;
      rf.B = 1; rf.H = 1; rf.W = (int)res->rows();
      o.res = &rf;
    }
    if (dst) {
      df = *dst;
      df.B = 1; df.H = 1; df.W = (int)dst->rows();
      o.dst = &df;
    }
    return x3_linear_ok(xf, N, 1, 1, 0, o);
  }

  // GroupNorm -> [FiLM: scale/shift rows of t_ss at column ss_col] -> SiLU as its own pass (layers the fused conv
  // does not take)
  T gn_silu(const T& x, const std::string& prefix, int ss_col) {
    const float* gamma = P(prefix + ".weight", x.C);
    const float* beta = P(prefix + ".bias", x.C);
    int G = cfg.resnet_groups;
    T y = alloc(x.B, x.H, x.W, x.C);
    emit_gn_stats(x, gamma, beta, ss_col, nullptr);
    size_t xo = x.at(), yo = y.off, so = gn_stats_t.off, sso = t_ss.off;
    int Bx = x.B, HW = x.HW(), C = x.C, ld = tmlp_total, ldx = x.LD();
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      const float* ssp = ss_col >= 0 ? uu->P(sso) + ss_col : nullptr;
      return launch_gn_apply_silu(uu->P(xo), ldx, uu->P(so), gamma, beta, ssp, ld, uu->P(yo), Bx, HW, C, G, s);
    }, "gn apply HW" + std::to_string(HW) + " C" + std::to_string(C));
    return y;
  }

  // ---- attention similarity variants (cfg.attn_qk_norm, include/kd_engine.h)
  float attn_scale() const {
    return cfg.attn_qk_norm == 1 ? 16.0f : cfg.attn_qk_norm == 2 ? 8.0f : 1.0f / sqrtf((float)cfg.attn_dim_head);
  }
  // in place on a workspace tensor: normalise (and scale) the `heads` 64-wide segments at the start of each row
  void qk_norm(const T& t, int ld, int heads, const float* scale_vec) {
    size_t off = t.at();   // (a column slice of a wider buffer: the fused q / kv projection)
    int64_t rows = t.rows();
    kd_unet* uu = u;
    emit([=](hipStream_t s) { return launch_l2norm_heads(uu->P(off), ld, rows, heads, scale_vec, s); },
         "qk l2norm rows" + std::to_string(rows) + " heads" + std::to_string(heads));
  }
  const float* q_scale_of(const std::string& pre) { return cfg.attn_qk_norm == 2 ? P(pre + ".q_scale", cfg.attn_dim_head) : nullptr; }
  const float* k_scale_of(const std::string& pre) { return cfg.attn_qk_norm == 2 ? P(pre + ".k_scale", cfg.attn_dim_head) : nullptr; }
  // learned null key / value [2][D]; with qk-norm the key row is normalised (and scaled) once, at plan build
  const float* null_kv_of(const std::string& pre) {
    const int D = cfg.attn_dim_head;
    const float* raw_nkv = P(pre + ".null_kv", 2 * D);
    if (!cfg.attn_qk_norm) return raw_nkv;
    const float* ks = k_scale_of(pre);
    return cached("null_kv_qknorm" + std::to_string(cfg.attn_qk_norm) + ":" + pre, (size_t)2 * D, [&](float* dst) {
      KD_HIP_THROW(hipMemcpyAsync(dst, raw_nkv, (size_t)2 * D * sizeof(float), hipMemcpyDeviceToDevice, 0));
      KD_THROW_IF(launch_l2norm_heads(dst, 2 * D, 1, 1, ks, 0));
    });
  }

  // ---- modules
  // cross attention of feature tokens to the conditioning tokens c [B,Nc,cond_dim]; returns attn(x)+x
  T cross_attn(const T& x, const std::string& pre, const T& c) {
    int H = cfg.attn_heads, D = cfg.attn_dim_head, inner = H * D, dim = x.C;
    T xn = layernorm(x, P(pre + ".norm.g", dim), nullptr);
    T q = linear(xn, P(pre + ".to_q.weight", (int64_t)inner * dim), nullptr, inner);
    free(xn);
    // K / V of the conditioning tokens: a function of c alone (cond region, see to_cond)
    const bool was_cond = to_cond;
    to_cond = cond_hoist;
    T kv = linear(c, P(pre + ".to_kv.weight", (int64_t)2 * inner * c.C), nullptr, 2 * inner);
    if (cfg.attn_qk_norm) qk_norm(kv, 2 * inner, H, k_scale_of(pre));
    to_cond = was_cond;
    const float* nkv = null_kv_of(pre);
    if (cfg.attn_qk_norm) qk_norm(q, inner, H, q_scale_of(pre));
    T o = alloc(x.B, x.H, x.W, inner);
    {
      size_t qo = q.off, kvo = kv.off, oo = o.off;
      int Bx = x.B, Nq = x.HW(), Nc = c.HW();
      float scale = attn_scale();
      kd_unet* uu = u;
      emit([=](hipStream_t s) {
        KVSeg s0{uu->P(kvo), uu->P(kvo) + inner, 2 * inner, Nc};
        KVSeg s1{nullptr, nullptr, 0, 0};
        return launch_attention(uu->P(qo), inner, nkv, nkv + D, s0, s1, uu->P(oo), inner, Bx, Nq, H, H, scale, s);
      }, "xattn Nq" + std::to_string(Nq) + " Nk" + std::to_string(Nc + 1));
      u->macs += (int64_t)Bx * H * Nq * (Nc + 1) * D * 2;
    }
    free(q);
    free(kv);
    T proj = linear(o, P(pre + ".to_out.0.weight", (int64_t)dim * inner), nullptr, dim);
    free(o);
    T y = layernorm(proj, P(pre + ".to_out.1.g", dim), nullptr, &x, ACT_NONE, nullptr, nullptr, true);   // block2's GroupNorm reads it
    free(proj);
    return y;
  }

  // TransformerBlock (depth 1): x = attn(x, ctx) + x ; x = ff(x) + x.
  // plain: the residual attention of earlier library versions (cfg.mid_attn_plain): parameters `<pre>.fn.fn.*`,
  // x = attn(x) + x and no feed-forward
  T transformer(const T& x, const std::string& pre, const T* ctx, const T* dst = nullptr, bool plain = false) {
    int H = cfg.attn_heads, D = cfg.attn_dim_head, inner = H * D, dim = x.C;
    std::string a = plain ? pre + ".fn.fn" : pre + ".layers.0.0", f = pre + ".layers.0.1";
    const bool qkv1 = cfg.conv_algo == 0 && kd_switch("KD_QKV_FUSED", 1) != 0;
    // (one reader - the fused q / k / v GEMM: where that runs on the bf16x3 kernel the LayerNorm leaves its planes)
    T xn = layernorm(x, P(a + ".norm.g", dim), nullptr, nullptr, ACT_NONE, nullptr, nullptr, false,
                     qkv1 && linear_is_x3(x, inner + 2 * D) ? 1 : 0);
    // to_q and to_kv read the same rows: one GEMM over the stacked weights [inner + 2 D][dim], q and k / v are column
    // slices of its output (the attention kernel takes row strides) - one launch less and fuller tiles
    const float* wq = P(a + ".to_q.weight", (int64_t)inner * dim);
    const float* wkv = P(a + ".to_kv.weight", (int64_t)2 * D * dim);
    T q, kv, qkv;
    if (qkv1) {
      const float* wqkv = cached("qkv:" + a, (size_t)(inner + 2 * D) * dim, [&](float* dst) {
        KD_HIP_THROW(hipMemcpyAsync(dst, wq, (size_t)inner * dim * sizeof(float), hipMemcpyDeviceToDevice, 0));
        KD_HIP_THROW(hipMemcpyAsync(dst + (size_t)inner * dim, wkv, (size_t)2 * D * dim * sizeof(float), hipMemcpyDeviceToDevice, 0));
      });
      qkv = linear(xn, wqkv, nullptr, inner + 2 * D);
      q = qkv;
      q.C = inner; q.ld = inner + 2 * D; q.coff = 0;
      kv = qkv;
      kv.C = 2 * D; kv.ld = inner + 2 * D; kv.coff = inner;
    } else {
      q = linear(xn, wq, nullptr, inner);
      kv = linear(xn, wkv, nullptr, 2 * D);
    }
    free(xn);
    T ckv;
    bool has_ctx = ctx != nullptr;
    if (has_ctx) {   // K / V of the context tokens: a function of c alone (cond region)
      const bool was_cond = to_cond;
      to_cond = cond_hoist;
      T cn = layernorm(*ctx, P(a + ".to_context.0.weight", ctx->C), P(a + ".to_context.0.bias", ctx->C));
      ckv = linear(cn, P(a + ".to_context.1.weight", (int64_t)2 * D * ctx->C), P(a + ".to_context.1.bias", 2 * D),
                   2 * D);
      free(cn);
      if (cfg.attn_qk_norm) qk_norm(ckv, 2 * D, 1, k_scale_of(a));
      to_cond = was_cond;
    }
    const float* nkv = null_kv_of(a);
    if (cfg.attn_qk_norm) {
      qk_norm(q, q.LD(), H, q_scale_of(a));
      qk_norm(kv, kv.LD(), 1, k_scale_of(a));
    }
    T o = alloc(x.B, x.H, x.W, inner);
    {
      size_t qo = q.at(), kvo = kv.at(), oo = o.off, co = has_ctx ? ckv.off : 0;
      int Bx = x.B, N = x.HW(), Nc = has_ctx ? ctx->HW() : 0;
      const int ldq = q.LD(), ldkv = kv.LD();
      float scale = attn_scale();
      kd_unet* uu = u;
      emit([=](hipStream_t s) {
        KVSeg s0{nullptr, nullptr, 0, 0};
        if (has_ctx) s0 = KVSeg{uu->P(co), uu->P(co) + D, 2 * D, Nc};
        KVSeg s1{uu->P(kvo), uu->P(kvo) + D, ldkv, N};
        return launch_attention(uu->P(qo), ldq, nkv, nkv + D, s0, s1, uu->P(oo), inner, Bx, N, H, 1, scale, s);
      }, "attn N" + std::to_string(N));
      u->macs += (int64_t)Bx * H * N * (N + Nc + 1) * D * 2;
    }
    if (qkv1) {
      free(qkv);
    } else {
      free(q);
      free(kv);
    }
    if (has_ctx) free(ckv);
    T proj = linear(o, P(a + ".to_out.0.weight", (int64_t)dim * inner), nullptr, dim);
    free(o);
    if (plain) {
      T x1 = layernorm(proj, P(a + ".to_out.1.g", dim), nullptr, &x);
      free(proj);
      return x1;
    }
    // x1 = LN(proj) g + x and the feed-forward's first LayerNorm h0 = LN(x1) g' in one pass over the rows
    T h0;
    int hidden = dim * cfg.ff_mult_x2 / 2;
    const bool gelu_late = linear_is_x3(proj, hidden);   // (h0 has proj's shape)
    T x1 = layernorm(proj, P(a + ".to_out.1.g", dim), nullptr, &x, ACT_NONE, P(f + ".0.g", dim), &h0, false, gelu_late ? 2 : 0);
    free(proj);
    // feed forward: Linear -> GELU -> LayerNorm -> Linear.  Where the first GEMM runs on the bf16x3 kernel (no activation
    // in its epilogue) it stores the raw product and the LayerNorm applies the GELU on the way in: the same function on the
    // same values
    T h1 = linear(h0, P(f + ".1.weight", (int64_t)hidden * dim), nullptr, hidden, gelu_late ? ACT_NONE : ACT_GELU);
    free(h0);
    T h2 = layernorm(h1, P(f + ".3.g", hidden), nullptr, nullptr, gelu_late ? ACT_GELU : ACT_NONE, nullptr, nullptr, false,
                     linear_is_x3(h1, dim, &x1, dst) ? 1 : 0);
    free(h1);
    T y = linear(h2, P(f + ".4.weight", (int64_t)dim * hidden), nullptr, dim, ACT_NONE, &x1, dst);
    free(h2);
    free(x1);
    return y;
  }

  // GlobalContext gate [B,1,1,C]
  T gca(const T& h, const std::string& pre) {
    int C = h.C, hid = std::max(3, C / 2);
    const float* wk = P(pre + ".to_k.weight", C);
    const float* bk = P(pre + ".to_k.bias", 1);
    const float* w0 = P(pre + ".net.0.weight", (int64_t)hid * C);
    const float* b0 = P(pre + ".net.0.bias", hid);
    const float* w2 = P(pre + ".net.2.weight", (int64_t)C * hid);
    const float* b2 = P(pre + ".net.2.bias", C);
    int Bx = h.B, HW = h.HW();
    kd_unet* uu = u;
    if (gca_gate_fused_ok(C, hid) && h.B > 1 && kd_switch("KD_GCA_FUSED", 1)) {   // (a batch-1 patch: one workgroup would stream the weights alone)
      T scratch = alloc_bytes(gca_scratch_floats(h.B, h.HW(), C) * sizeof(float));
      T gate = alloc(h.B, 1, 1, C);
      size_t ho = h.off, so = scratch.off, go = gate.off;
      emit([=](hipStream_t s) {
        return launch_gca_gate(uu->P(ho), wk, bk, uu->P(so), w0, b0, hid, w2, b2, uu->P(go), Bx, HW, C, s);
      }, "gca_gate HW" + std::to_string(HW) + " C" + std::to_string(C));
      u->macs += (int64_t)Bx * HW * C * 2 + (int64_t)Bx * C * hid * 2;
      free(scratch);
      return gate;
    }
    T logits = alloc(h.B, h.H, h.W, 1);
    T pooled = alloc(h.B, 1, 1, C);
    T scratch = alloc_bytes(gca_scratch_floats(h.B, h.HW(), C) * sizeof(float));
    size_t ho = h.off, lo = logits.off, po = pooled.off, so = scratch.off;
    emit([=](hipStream_t s) {
      return launch_gca_pool(uu->P(ho), wk, bk, uu->P(lo), uu->P(po), uu->P(so), Bx, HW, C, s);
    }, "gca_pool HW" + std::to_string(HW) + " C" + std::to_string(C));
    u->macs += (int64_t)Bx * HW * C * 2;
    free(logits);
    free(scratch);
    T hidden = alloc(h.B, 1, 1, hid);
    skinny(pooled.off, C, w0, b0, hidden.off, hid, h.B, C, hid, ACT_NONE, ACT_SILU);
    free(pooled);
    T gate = alloc(h.B, 1, 1, C);
    skinny(hidden.off, hid, w2, b2, gate.off, C, h.B, hid, C, ACT_NONE, ACT_SIGMOID);
    free(hidden);
    return gate;
  }

  // ---- Winograd F(2x2,3x3) for the `Block` = GroupNorm -> [FiLM] -> SiLU -> conv3x3 of deep layers
  // (kernels_wino.hip).  Worth it where the 2.25x smaller GEMM outweighs moving 4x the map through
  // the transforms: measured on MI355X, Cin >= 256 wins and Cin = 128 loses (profiles/README.md).
  // cfg.conv_algo: 0 auto, 1 never, >= 32 explicit Cin threshold (experiments, tests).
  bool wino_ok(const T& x, int cout) const {
    if (cfg.conv_algo == 1 || cfg.conv_algo == 4) return false;
    const int min_cin = cfg.conv_algo >= 32 ? cfg.conv_algo : 256;
    // the fused kernel (fwino_ok) takes every layer whose map it can tile and whose launch fills the chip, up to the
    // Cin its affine table holds.  Measured on the 64->256 UNet, ms/step with the hand-over at Cin <= 0 / 256 / 512 /
    // 1024 / all: 48.1 / 44.8 / 44.5 / 45.1 / 45.3 with the eight-wave kernel of round 1 (its Cin = 1024 launches lost
    // to the batched GEMMs), 37.59 / 37.36 / 37.40 at <= 512 / 1024 / 2048 with the persistent sixteen-wave kernel
    // (300 us against 265 + 40 + 21 for GEMM + transforms).  KD_FWINO_MAX_CIN moves it for experiments, read per plan
    const int fw_max = kd_switch("KD_FWINO_MAX_CIN", 2048);
    if (wino4_ok(x, cout)) return false;
    if (cfg.conv_algo < 32 && x.C <= fw_max && fwino_ok(x, cout)) return false;
    if ((x.H & 1) || (x.W & 1) || x.C < min_cin || x.C % 32 || cout <= 32 || cout % 4) return false;
    const int64_t Mt = (int64_t)x.B * (x.H / 2) * (x.W / 2);
    return Mt % 256 == 0 && 16 * Mt < 0x7fffffff && (int64_t)16 * cout * x.C * 4 < 0x7fffffff;
  }
  // ---- Winograd F(4x4,3x3) (kernels_wino4.hip): 36 batched GEMMs over tiles of 4x4 outputs, 4x fewer MFMA issues than
  // the direct conv and 1.78x fewer than F(2x2,3x3), for V / D transform buffers of 2.25x the map.  cfg.conv_algo 0:
  // layers with Cin >= cfg.wino43_min_cin (default 512) whose GEMMs fill the chip - measured against the fused
  // F(2x2,3x3) kernel at batch 16 (profiles/README.md); 4: wherever the shape fits (tests); 1 / 2 / 3 / >= 32: never.
  // images per launch set of a F(4x4,3x3) layer: the whole batch where the rule below takes it, otherwise (default plan with
  // the bf16x3 GEMMs) the largest divisor of the batch whose V / D / maps stay inside what a buffer resource spans (4 GB) -
  // unet3's 512 x 512 and 1024 x 1024 levels at batch 8, which otherwise fall back to the fused F(2x2,3x3) kernel; 0 = not a
  // F(4x4,3x3) layer
  int wino4_images(const T& x, int cout) const {
    const int cap = cfg.wino4_max_images > 0 ? cfg.wino4_max_images : x.B;
    if (cap >= x.B && wino4_whole_ok(x, cout)) return x.B;
    if (cap >= x.B && (cfg.conv_algo != 0 || cfg.wino43_min_cin != 0 || cfg.gemm_bf16x3 < 0 || to_text || to_static)) return 0;
    if (cap < x.B && !wino4_whole_ok(x, cout)) return 0;   // (the test knob cuts layers the plan takes, nothing else)
    for (int ns = 2; ns <= x.B; ++ns) {
      if (x.B % ns) continue;
      T xs = x;
      xs.B = x.B / ns;
      if (xs.B > cap || (int64_t)xs.B * x.H * x.W * x.LD() * 4 >= ((int64_t)1 << 32)) continue;
      const int64_t Mt = (int64_t)xs.B * (x.H / 4) * (x.W / 4);
      if (cap >= x.B && ((x.H & 3) || (x.W & 3) || !gemm_bf16x3_ok(36, Mt, cout, x.C))) continue;
      if (wino4_whole_ok(xs, cout)) return xs.B;
    }
    return 0;
  }
  bool wino4_ok(const T& x, int cout) const { return wino4_images(x, cout) > 0; }
  bool wino4_whole_ok(const T& x, int cout) const {
    if (cfg.conv_algo != 0 && cfg.conv_algo != 4) return false;
    if (cfg.wino43_min_cin < 0 && cfg.conv_algo == 0) return false;
    if ((x.H & 3) || (x.W & 3) || x.C % 32 || cout % 64 || cout < 64) return false;
    const int64_t Mt = (int64_t)x.B * (x.H / 4) * (x.W / 4);
    if (Mt % 128 || 36 * Mt >= 0x7fffffff || (int64_t)36 * cout * x.C * 4 >= 0x7fffffff) return false;
    if (cfg.conv_algo == 4) return true;
    if ((36 * Mt / 128) * (cout / 64) < 512) return false;   // the 36 GEMMs must fill the chip
    // default threshold Cin >= 512.  Below it the transform passes (6.5 x the map through HBM) eat the GEMM saving:
    // same-box A/B against the fused F(2x2,3x3) kernel at batch 16: 256 -> 256 at 64 x 64 299 -> 286 us (-4 %, not
    // worth six times the per-conv rounding error), 256 -> 128 at 128 x 128 626 -> 737 (+18 %), at 256 x 256 +17 %
    // (round 4, with the input transform 13 % faster: Cin = 256 on the 64 x 64 level alone, 9 layers, gives 32.74 -> 32.59 ms
    // per forward - 0.5 % for six times those layers' rounding error: the threshold stays at 512)
    // With the position GEMMs on the bf16 pipe (bf16x3, 1.5 x the fp32 MFMA rate) the hand-over moves down to Cin >= 256
    // on maps of up to 65536 pixels - same box, per layer: 256 -> 256 at 64 x 64 (batch 16) 330 us fused F(2x2,3x3) against
    // 129 + 82 + 48 (GEMM + transforms), nine layers, -0.64 ms per forward; the larger maps stay: 256 -> 128 at 128 x 128
    // 688 against 256 + 378 + ~100, at 256 x 256 2618 against 934 + 1514 + ~400 (the transforms write 3.4 x the map)
    int thr = cfg.wino43_min_cin > 0 ? cfg.wino43_min_cin : 512;
    // (round 5: 256 -> 256 on larger maps as well - unet3's 256 x 256 level at batch 8, 524288 pixels: 2.62 ms fused against
    // 8 x (117 + 78 + 49 + 12) us for the same pixels at batch 1; what loses above 65536 pixels is Cout = 128, whose GEMMs
    // are half as wide for the same transform traffic)
    // (later in round 5, with the input transform's loads issued together - 185 -> 121 us at 65536 pixels x 512 channels -
    // Cout = 128 wins as well: 256 -> 128 at 128 x 128 698 us fused against 241 + 258 + 80, at 256 x 256 2639 against 1031 + 934
    // + 299; same-box bench 27.81 / 27.89 -> 27.56 / 27.62 ms: the rule is Cin >= 256 wherever the bf16x3 GEMM takes the shape)
    // (end of round 5: Cin = 128 too, with V of those HBM-bound layers kept in fp32 - wino4_block -: 128 -> 128 at 256 x 256
    // 1509 us fused against 332 + 608 + 333, at 128 x 128 375 against 88 + 166 + 88)
    if (cfg.wino43_min_cin == 0 && cfg.gemm_bf16x3 >= 0 && gemm_bf16x3_ok(36, Mt, cout, x.C)) thr = 128;
    return x.C >= thr;
  }
  // skip_c0 >= 0: channels [skip_c0, ..) of x hold an unscaled skip tensor the layer must see times skip_scale: the input
  // transform folds the factor into its affine (the statistics come from partials that carry the scale: add_skip)
  T wino4_block(const T& x, const std::string& gn_prefix, int ss_col, const std::string& conv_prefix, int Cout,
                const T* res, int skip_c0 = -1, float skip_scale = 1.0f) {
    const int Cin = x.C, G = cfg.resnet_groups, H = x.H, W = x.W, HW = x.HW();
    // Bx images per set of launches (wino4_images): V and D are one set's, the sets run one after the other
    const int Bs_ = wino4_images(x, Cout), Bx = Bs_ > 0 ? Bs_ : x.B, nset = x.B / Bx;
    const int64_t Mt = (int64_t)Bx * (H / 4) * (W / 4);
    const float* gamma = P(gn_prefix + ".weight", Cin);
    const float* beta = P(gn_prefix + ".bias", Cin);
    const float* bias = P(conv_prefix + ".bias", Cout);
    const float* wsrc = raw(conv_prefix + ".weight", (int64_t)Cout * Cin * 9);
    float* U = cached("wino4:" + conv_prefix, (size_t)36 * Cout * Cin,
                      [&](float* dst) { KD_THROW_IF(launch_wino4_pack(wsrc, dst, Cout, Cin, 0)); });
    emit_gn_stats(x, gamma, beta, ss_col, nullptr);
    // the 36 GEMMs as fp32-class products on the bf16 matrix pipe (kernels_gemm_bf16x3.hip) where the shape fits its tile
    const bool x3 = cfg.gemm_bf16x3 >= 0 && !to_text && !to_static && gemm_bf16x3_ok(36, Mt, Cout, Cin);
    // V written as planes by the input transform, or as fp32, split by the GEMM's loader waves on the way into LDS: the
    // transform then writes a third less and is that much faster, but the loaders' vector work beside the MFMA waves costs a
    // GEMM that is bound by the matrix pipe 10-15 %.  Per layer (same box, batch 16, transform + GEMM in us, planes / fp32;
    // with the loader waves' loads three stages in flight, kernels_gemm_bf16x3.hip):
    //   128 -> 128 at 256 x 256 480 + 568 / 336 + 526    256 -> 128 at 256 x 256 1028 + 935 / 671 + 858
    //   128 -> 128 at 128 x 128 118 + 168 / 90 + 146     256 -> 128 at 128 x 128 241 + 256 / 164 + 238
    //   256 -> 256 at 64 x 64 65.5 + 112 / 50.5 + 121    512 -> 256 at 64 x 64 123 + 190 / 90 + 215
    //   512 -> 512 at 32 x 32 28.5 + 103 / 25.5 + 115    1024 -> 1024 at 16 x 16 19.6 + 102 / 16.8 + 113
    // fp32 wins where the GEMM itself waits for HBM - few MACs per byte of V and D: Cin Cout / (6 Cin + 4 Cout) < 40 (12.8 ...
    // 32 for the rows that win, 51 upwards for those that lose).  cfg.gemm_bf16x3: 0 = by that rule, 1 = planes, 2 = fp32
    const bool v_f32 = cfg.gemm_bf16x3 == 2 || (cfg.gemm_bf16x3 == 0 && (int64_t)Cin * Cout < 40 * (6 * (int64_t)Cin + 4 * Cout));
    const bool x3_planes = x3 && !v_f32;
    T V = x3_planes ? alloc_bytes((size_t)36 * Mt * Cin * 6) : alloc(1, 1, (int)(36 * Mt), Cin);
    T D = alloc(1, 1, (int)(36 * Mt), Cout);
    T y = alloc(x.B, H, W, Cout);
    const std::string shape = " M" + std::to_string((int64_t)Bx * HW) + " Cin" + std::to_string(Cin) + " Cout" +
                              std::to_string(Cout);
    kd_unet* uu = u;
    const bool sg = seg_on && Cout % 64 == 0;   // GroupNorm partials of y for whichever layer normalises it next
    const size_t sgo_all = sg ? add_seg(y, 0, Cout / 16, (H / 4) * (W / 4)) : 0;
    for (int st = 0; st < nset; ++st) {
      const int b0 = st * Bx;   // first image of the set
      {
        size_t xo = x.at() + (size_t)b0 * HW * x.LD() * sizeof(float), vo = V.off;
        size_t so = gn_stats_t.off + (size_t)b0 * G * 2 * sizeof(float), sso = t_ss.off + (size_t)b0 * tmlp_total * sizeof(float);
        const int ld = tmlp_total, ldx = x.LD();
        emit([=](hipStream_t s) {
          const float* ssp = ss_col >= 0 ? uu->P(sso) + ss_col : nullptr;
          if (x3_planes)
            return launch_wino4_in3(uu->P(xo), ldx, uu->P(so), gamma, beta, ssp, ld, uu->P(vo), Bx, H, W, Cin, G, s, skip_c0, skip_scale);
          return launch_wino4_in(uu->P(xo), ldx, uu->P(so), gamma, beta, ssp, ld, uu->P(vo), Bx, H, W, Cin, G, s, skip_c0, skip_scale);
        }, (x3_planes ? "wino4_in3" : "wino4_in") + shape);
      }
      if (x3) {
        const float* U3 = cached("wino4x3:" + conv_prefix, ((size_t)36 * Cout * Cin * 3 + 1) / 2,
                                 [&](float* dst) { KD_THROW_IF(launch_split3(U, dst, 36, Cout, Cin, 0)); });
        const size_t vo = V.off, d_o = D.off;
        const int64_t macs = (int64_t)Bx * HW * Cout * Cin * 9;   // the algorithmic MACs of the 3x3 conv it replaces
        if (!u->x3_ws) KD_HIP_THROW(hipMalloc(&u->x3_ws, gemm_bf16x3_workspace_bytes()));
        emit([=](hipStream_t s) { return launch_gemm_bf16x3(uu->P(vo), U3, uu->P(d_o), 36, (int)Mt, Cout, Cin, uu->x3_ws, s, !x3_planes, false); },
             "wino4 gemm bf16x3" + shape, macs);
        u->macs += macs;
        u->op_mfma.back() = 6 * 36 * Mt * Cout * Cin;   // bf16 MACs
        u->mfma_bf16_macs += u->op_mfma.back();
        if (gemm_bf16x3_needs_sum(36, (int)Mt, Cout, Cin))   // the left-over tiles' k-parts (its own launch: the whole chip adds them)
          emit([=](hipStream_t s) { return launch_gemm_bf16x3_sum(uu->P(d_o), 36, (int)Mt, Cout, Cin, uu->x3_ws, s); },
               "wino4 x3 sum" + shape);
      } else {
        ConvOpt o;
        o.wz_rows = (int)Mt;
        o.wz_count = 36;
        o.dst = &D;
        o.macs_override = (int64_t)Bx * HW * Cout * Cin * 9;   // the algorithmic MACs of the 3x3 conv it replaces
        conv(V, U, nullptr, Cout, 1, 1, 0, o);
        if (!to_text && !to_static) u->op_label.back() = "wino4 gemm" + shape;
      }
      {
        const size_t sgo = sgo_all + (size_t)b0 * (Cout / 16) * ((H / 4) * (W / 4)) * 2 * sizeof(double);
        size_t d_o = D.off, yo = y.off + (size_t)b0 * HW * Cout * sizeof(float);
        const bool hr = res != nullptr;
        const int ldres = res ? res->LD() : 0;
        size_t ro = res ? res->at() + (size_t)b0 * HW * ldres * sizeof(float) : 0;
        emit([=](hipStream_t s) {
          return launch_wino4_out(uu->P(d_o), bias, hr ? uu->P(ro) : nullptr, ldres, uu->P(yo), Cout,
                                  sg ? (double*)uu->P(sgo) : nullptr, Bx, H, W, Cout, s);
        }, "wino4_out" + shape);
      }
    }   // sets
    free(V);
    free(D);
    return y;
  }

  T wino_block(const T& x, const std::string& gn_prefix, int ss_col, const std::string& conv_prefix, int Cout,
               const T* res) {
    const int Cin = x.C, G = cfg.resnet_groups, Bx = x.B, H = x.H, W = x.W, HW = x.HW();
    const int64_t Mt = (int64_t)Bx * (H / 2) * (W / 2);
    const float* gamma = P(gn_prefix + ".weight", Cin);
    const float* beta = P(gn_prefix + ".bias", Cin);
    const float* bias = P(conv_prefix + ".bias", Cout);
    const float* wsrc = raw(conv_prefix + ".weight", (int64_t)Cout * Cin * 9);
    float* U = cached("wino:" + conv_prefix, (size_t)16 * Cout * Cin,
                      [&](float* dst) { KD_THROW_IF(launch_wino_pack(wsrc, dst, Cout, Cin, 0)); });
    // The map can be walked in slices of tiles (V and D are 4x the slice each) to bound the workspace:
    // cfg.wino_slice_mb caps V+D per slice.  Default: one slice - slices small enough to stay in the
    // 256 MB Infinity Cache were measured and are slower (56.5 ms/step unsliced, 59.3 at 96 MB,
    // 57.5 at 192 MB): the cache does not turn the V/D round trip into hits.
    const int64_t slice_mb = cfg.wino_slice_mb;
    int64_t nt_slice = Mt;
    if (slice_mb > 0) {
      nt_slice = (slice_mb << 20) / (64 * (int64_t)(Cin + Cout)) / 256 * 256;
      nt_slice = std::min(Mt, std::max<int64_t>(nt_slice, 256));
    }
    emit_gn_stats(x, gamma, beta, ss_col, nullptr);
    T V = alloc(1, 1, (int)(16 * nt_slice), Cin);
    T D = alloc(1, 1, (int)(16 * nt_slice), Cout);
    T y = alloc(Bx, H, W, Cout);
    // GroupNorm partials of y from the output transform (one chunk per 2x2 tile) for whichever layer normalises it next
    const bool sg = seg_on && Cout % 16 == 0;
    const size_t sgo = sg ? add_seg(y, 0, Cout / 16, (H / 2) * (W / 2)) : 0;
    const std::string shape = " M" + std::to_string((int64_t)Bx * HW) + " Cin" + std::to_string(Cin) + " Cout" +
                              std::to_string(Cout);
    for (int64_t t0 = 0; t0 < Mt; t0 += nt_slice) {
      const int64_t nt = std::min(nt_slice, Mt - t0);
      {
        size_t xo = x.at(), vo = V.off, so = gn_stats_t.off, sso = t_ss.off;
        int ld = tmlp_total, ldx = x.LD();
        kd_unet* uu = u;
        emit([=](hipStream_t s) {
          const float* ssp = ss_col >= 0 ? uu->P(sso) + ss_col : nullptr;
          return launch_wino_in(uu->P(xo), ldx, uu->P(so), gamma, beta, ssp, ld, uu->P(vo), Bx, H, W, Cin, G, t0, nt,
                                s);
        }, "wino_in" + shape);
      }
      T Vs = V, Ds = D;
      Vs.W = (int)(16 * nt);
      Ds.W = (int)(16 * nt);
      ConvOpt o;
      o.wz_rows = (int)nt;
      o.dst = &Ds;
      o.macs_override = nt * 4 * Cout * Cin * 9;  // algorithmic MACs of the share of the 3x3 conv it replaces
      conv(Vs, U, nullptr, Cout, 1, 1, 0, o);
      if (!to_text && !to_static) u->op_label.back() = "wino gemm" + shape;
      {
        size_t d_o = D.off, yo = y.off, ro = res ? res->at() : 0;
        bool hr = res != nullptr;
        int ldres = res ? res->LD() : 0;
        kd_unet* uu = u;
        emit([=](hipStream_t s) {
          return launch_wino_out(uu->P(d_o), bias, hr ? uu->P(ro) : nullptr, ldres, uu->P(yo), Bx, H, W, Cout, t0, nt,
                                 s, sg ? (double*)uu->P(sgo) : nullptr);
        }, "wino_out" + shape);
      }
    }
    free(V);
    free(D);
    return y;
  }

  // ---- fused Winograd F(2x2,3x3) + GroupNorm / FiLM / SiLU (kernels_wino_fused128.hip): every ResnetBlock 3x3 conv
  // whose map the kernel can tile (H % 8, W % 16, Cout % 128: all reference configs; a reduced-width model's narrow
  // levels take the batched-GEMM or the direct path).  cfg.conv_algo 0: wherever the shape fits and the launch fills
  // the chip; 1 and 2: never (2 = the batched-GEMM Winograd path only); 3: wherever the shape fits (tests).
  bool fwino_ok(const T& x, int cout) const {
    if (cfg.conv_algo == 1 || cfg.conv_algo == 2) return false;
    if (x.C < 32 || !wino_fused128_ok(x.B, x.H, x.W, x.C, cout)) return false;
    if (cfg.conv_algo == 3) return true;
    return (int64_t)x.B * (x.H / 16) * (x.W / 16) * (cout / 64) >= 256;  // one workgroup per CU and round
  }
  // The same layer straight from the un-normalised block input: GroupNorm statistics, the per-(image, channel)
  // affine fold, and the fused kernel that applies GroupNorm / FiLM / SiLU to the raw patch in LDS
  // (wino_fused_gn_kernel): the activated map is never written.  KD_FWINO_GN=0 keeps the separate
  // gn_apply_silu pass (A/B, read per plan).
  bool fwino_gn_ok(const T& x, int cout) const {
    const bool on = kd_switch("KD_FWINO_GN", 1) != 0;
    return on && x.C <= wino_fused_gn_max_cin() && x.C % cfg.resnet_groups == 0 && fwino_ok(x, cout) &&
           (int64_t)x.H * x.W * x.LD() * 4 < 0x7fffffff;
  }
  T fwino_gn_conv(const T& x, const std::string& gn_prefix, int ss_col, const std::string& conv_prefix, int Cout,
                  const T* res) {
    const int Cin = x.C, G = cfg.resnet_groups, Bx = x.B, H = x.H, W = x.W;
    const float* gamma = P(gn_prefix + ".weight", Cin);
    const float* beta = P(gn_prefix + ".bias", Cin);
    const float* bias = P(conv_prefix + ".bias", Cout);
    const float* wsrc = raw(conv_prefix + ".weight", (int64_t)Cout * Cin * 9);
    float* U = cached("winof_gn128:" + conv_prefix, (size_t)16 * Cout * Cin,
                      [&](float* dst) { KD_THROW_IF(launch_wino_fused128_pack(wsrc, dst, Cout, Cin, 0, WF_U_SCALE)); });
    T ab = alloc_bytes((size_t)Bx * Cin * 2 * sizeof(float));
    T y = alloc(Bx, H, W, Cout);
    kd_unet* uu = u;
    // statistics: from the partials x's producer left (one launch gives the folded affine too), else a pass over x
    if (!emit_gn_stats(x, gamma, beta, ss_col, &ab)) {
      size_t so = gn_stats_t.off, sso = t_ss.off, abo = ab.off;
      const int ld = tmlp_total;
      emit([=](hipStream_t s) {
        const float* ssp = ss_col >= 0 ? uu->P(sso) + ss_col : nullptr;
        return launch_gn_fold(uu->P(so), gamma, beta, ssp, ld, uu->P(abo), Bx, Cin, G, s);
      }, "gn fold C" + std::to_string(Cin));
    }
    // the epilogue leaves the partials of y for whichever GroupNorm reads it next (block2, or the next block)
    const bool so_ = seg_on && Cout % 16 == 0 && kd_switch("KD_FWINO_STATS", 1) != 0;
    const size_t pout = so_ ? add_seg(y, 0, Cout / 16, (int)wino_fused_out_stats_chunks(H, W, Cout, Cout / 16)) : 0;
    size_t xo = x.at(), yo = y.off, ro = res ? res->at() : 0, abo = ab.off;
    const bool hr = res != nullptr;
    const int ldres = res ? res->LD() : 0, ldx = x.LD();
    const int64_t m = (int64_t)Bx * H * W * Cout * Cin * 9;
    // id -> (image, y0, x0, slab) of the kernel's work items: one table per map shape, shared by the layers
    const std::string shape_key = std::to_string(Bx) + "x" + std::to_string(H) + "x" + std::to_string(W) + "x" + std::to_string(Cout);
    const float* items = cached("winof128_items:" + shape_key, wino_fused128_items_count(Bx, H, W, Cout) * 4,
                                [&](float* dst) { KD_THROW_IF(launch_wino_fused128_items(dst, Bx, H, W, Cout, 0)); });
    emit([=](hipStream_t s) {
      return launch_wino_fused_gn128(uu->P(xo), ldx, uu->P(abo), U, bias, hr ? uu->P(ro) : nullptr, ldres, uu->P(yo), Bx, H,
                                     W, Cin, Cout, so_ ? (double*)uu->P(pout) : nullptr, so_ ? Cout / 16 : 0, items, s);
    }, "wino fused M" + std::to_string((int64_t)Bx * H * W) + " Cin" + std::to_string(Cin) + " Cout" +
           std::to_string(Cout), m);
    free(ab);
    if (!to_text) u->macs += m;
    if (!to_text && !to_static) {
      const int64_t issued = (int64_t)Bx * H * W * 4 * Cout * ((Cin / 4 + 3) / 4 * 16);
      u->mfma_macs += issued;
      u->op_mfma.back() = issued;
    }
    return y;
  }

  // ResnetBlock.  Does NOT free x.
  // ct: the buffer of the skip concat that follows this block ([B,H,W,dim_out + skip channels]); when the
  // block ends in its 1x1 skip conv, that conv writes the block output straight into the first dim_out
  // channels of ct and ct is returned (concat_skip then only adds the skip half); otherwise ct is ignored
  // out_slot: a channel slice (of the concat buffer this block's output will later be part of as the skip half);
  // a block that ends in gate_add writes its output there and returns the slice; otherwise out_slot is ignored.
  // skip_c0 >= 0: x is a concat whose channels [skip_c0, ..) hold an UNSCALED skip tensor that the layer must see
  // scaled by skip_scale: the GroupNorm gets it through the partials' scale (SegPart), the 1x1 skip conv through
  // weights whose columns were scaled at plan build.
  T resnet(const T& x, const std::string& pre, int dim_out, const T* ctx, bool use_gca, const T* ct = nullptr,
           const T* out_slot = nullptr, int skip_c0 = -1, float skip_scale = 1.0f) {
    bool has_cross = has(pre + ".cross_attn.to_q.weight");
    if (has_cross && !ctx) throw std::runtime_error("cross-attention block without conditioning tokens: " + pre);
    int dim_in = x.C;
    T h;
    if (wino4_ok(x, dim_out)) {
      h = wino4_block(x, pre + ".block1.groupnorm", -1, pre + ".block1.project", dim_out, nullptr, skip_c0, skip_scale);
    } else if (wino_ok(x, dim_out)) {
      h = wino_block(x, pre + ".block1.groupnorm", -1, pre + ".block1.project", dim_out, nullptr);
    } else if (fwino_gn_ok(x, dim_out)) {
      h = fwino_gn_conv(x, pre + ".block1.groupnorm", -1, pre + ".block1.project", dim_out, nullptr);
    } else {
      T y1 = gn_silu(x, pre + ".block1.groupnorm", -1);
      ConvOpt o1;
      o1.want_seg = true;   // block2's GroupNorm (or the cross-attention's input has none: harmless) reads h next
      h = conv(y1, pack_conv(pre + ".block1.project.weight", dim_out, dim_in, dim_in, 3),
               P(pre + ".block1.project.bias", dim_out), dim_out, 3, 1, 1, o1);
      free(y1);
    }
    if (has_cross) {
      T h2 = cross_attn(h, pre + ".cross_attn", *ctx);
      free(h);
      h = h2;
    }
    auto it = tmlp_off.find(pre);
    int ss_col = it != tmlp_off.end() ? it->second : -1;
    bool has_res_conv = has(pre + ".res_conv.weight");
    T h2;
    if (wino4_ok(h, dim_out)) {
      h2 = wino4_block(h, pre + ".block2.groupnorm", ss_col, pre + ".block2.project", dim_out,
                       (!use_gca && !has_res_conv) ? &x : nullptr);
      free(h);
    } else if (wino_ok(h, dim_out)) {
      h2 = wino_block(h, pre + ".block2.groupnorm", ss_col, pre + ".block2.project", dim_out,
                      (!use_gca && !has_res_conv) ? &x : nullptr);
      free(h);
    } else if (fwino_gn_ok(h, dim_out)) {
      h2 = fwino_gn_conv(h, pre + ".block2.groupnorm", ss_col, pre + ".block2.project", dim_out,
                         (!use_gca && !has_res_conv) ? &x : nullptr);
      free(h);
    } else {
      T y2 = gn_silu(h, pre + ".block2.groupnorm", ss_col);
      ConvOpt o2;
      if (!use_gca && !has_res_conv) o2.res = &x;  // h2 + x folded into the conv epilogue
      o2.want_seg = true;
      free(h);
      h2 = conv(y2, pack_conv(pre + ".block2.project.weight", dim_out, dim_out, dim_out, 3),
                P(pre + ".block2.project.bias", dim_out), dim_out, 3, 1, 1, o2);
      free(y2);
    }
    if (!use_gca && !has_res_conv) return h2;
    if (skip_c0 >= 0 && !has_res_conv) throw std::runtime_error("plan: folded skip scale without a skip conv: " + pre);
    const float* w_res = nullptr;
    if (has_res_conv) {
      w_res = P(pre + ".res_conv.weight", (int64_t)dim_out * dim_in);
      if (skip_c0 >= 0) {   // columns of the skip channels times skip_scale
        const float* src = w_res;
        w_res = cached("res_conv_skipscaled:" + pre + ":" + std::to_string(skip_c0), (size_t)dim_out * dim_in, [&](float* dst) {
          KD_THROW_IF(launch_copy_scale_rows(src, dim_in, dst, dim_in, dim_in, 1.0f, dim_out, 0));
          KD_THROW_IF(launch_copy_scale_rows(dst + skip_c0, dim_in, dst + skip_c0, dim_in, dim_in - skip_c0, skip_scale,
                                             dim_out, 0));
        });
      }
    }
    T out;
    if (use_gca) {
      T gate = gca(h2, pre + ".gca");
      if (has_res_conv) {
        ConvOpt o;
        o.gate_src = &h2;
        o.gate = &gate;
        o.want_seg = true;
        if (ct) o.dst = ct;
        out = conv(x, w_res, P(pre + ".res_conv.bias", dim_out), dim_out, 1, 1, 0, o);
      } else {
        out = out_slot ? *out_slot : alloc(x.B, x.H, x.W, dim_out);
        if (out.C != dim_out) throw std::runtime_error("plan: output slot of the wrong width: " + pre);
        size_t ao = h2.off, go = gate.off, ro = x.at(), yo = out.at();
        int Bx = x.B, HW = x.HW(), ldr = x.LD(), ldy = out.LD();
        const bool sg = seg_on && dim_out % 16 == 0;   // GroupNorm partials of `out` for the block that reads it
        const size_t sgo = sg ? add_seg(out, 0, dim_out / 16, gate_add_chunks(Bx, HW)) : 0;
        kd_unet* uu = u;
        emit([=](hipStream_t s) {
          return launch_gate_add(uu->P(ao), uu->P(go), uu->P(ro), ldr, uu->P(yo), ldy,
                                 sg ? (double*)uu->P(sgo) : nullptr, Bx, HW, dim_out, s);
        }, "gate_add HW" + std::to_string(HW) + " C" + std::to_string(dim_out));
      }
      free(gate);
    } else {  // res_conv without gca: out = conv1x1(x) + h2
      ConvOpt o;
      o.res = &h2;
      o.want_seg = true;
      if (ct) o.dst = ct;
      out = conv(x, w_res, P(pre + ".res_conv.bias", dim_out), dim_out, 1, 1, 0, o);
    }
    free(h2);
    return out;
  }

  // the Downsample (pixel-unshuffle + conv1x1 = a 2 x 2 / stride-2 conv) on the bf16x3 kernel: its fp32-A loader gathers the
  // four input pixels of an output pixel (X3Epi::a_tap_c), K = 4 C
  X3Epi downsample_x3_epi(const T& x, int Cout) const {
    X3Epi e;
    e.lda = x.LD();
    e.ldy = Cout;
    e.a_tap_c = x.C;
    e.a_wi = x.W;
    e.a_hi = x.H;
    e.hw = (x.H / 2) * (x.W / 2);
    return e;
  }
  bool downsample_x3_ok(const T& x, int Cout) const {
    if (cfg.gemm_bf16x3 < 0 || cfg.conv_algo != 0 || cfg.x3_linear < 0 || to_text || to_static || to_cond) return false;
    if ((x.H & 1) || (x.W & 1) || x.C % 16 || Cout % 128) return false;
    const int K = 4 * x.C;
    const int64_t M = (int64_t)x.B * (x.H / 2) * (x.W / 2);
    if (K < (cfg.x3_linear > 0 ? cfg.x3_linear : 256) || M % 256 || (M / 256) * (Cout / 128) < 64) return false;
    if (cfg.x3_linear == 0 && K < 1024 && gemm_bf16x3_needs_sum(1, (int)M, Cout, K)) return false;
    return gemm_bf16x3_epi_ok(M, Cout, K, downsample_x3_epi(x, Cout));
  }
  T downsample_x3(const T& x, const std::string& pre, const float* w_taps /*[tap][O][C]*/, const float* bias, int Cout) {
    const int C = x.C, K = 4 * C, Ho = x.H / 2, Wo = x.W / 2;
    const int64_t M = (int64_t)x.B * Ho * Wo;
    // B operand [O][K] with k = tap C + c, then its three planes
    const float* W3 = cached("x3down:" + pre, ((size_t)Cout * K * 3 + 1) / 2, [&](float* dst) {
      float* wk = nullptr;
      KD_HIP_THROW(hipMalloc((void**)&wk, (size_t)Cout * K * sizeof(float)));
      int rc = 0;
      for (int t = 0; t < 4 && !rc; ++t)
        rc = launch_copy_scale_rows(w_taps + (size_t)t * Cout * C, C, wk + (size_t)t * C, K, C, 1.0f, Cout, 0);
      if (!rc) rc = launch_split3(wk, dst, 1, Cout, K, 0);
      hipError_t er = hipDeviceSynchronize();
      (void)hipFree(wk);
      KD_THROW_IF(rc);
      KD_HIP_THROW(er);
    });
    if (!u->x3_ws) KD_HIP_THROW(hipMalloc(&u->x3_ws, gemm_bf16x3_workspace_bytes()));
    T y = alloc(x.B, Ho, Wo, Cout);
    const X3Epi base = downsample_x3_epi(x, Cout);
    // the output feeds the GroupNorm of the level's first ResnetBlock: partials from the epilogue
    const bool sg = seg_on && Cout % 16 == 0 && (Ho * Wo) % 32 == 0;
    const int seg_rows = gemm_bf16x3_seg_rows((int)M, Cout, K);
    const size_t sgo = sg ? add_seg(y, 0, Cout / 16, Ho * Wo / seg_rows) : 0;
    const size_t xo = x.at(), yo = y.off;
    kd_unet* uu = u;
    auto epi_of = [=]() {
      X3Epi e = base;
      e.bias = bias;
      if (sg) {
        e.seg = (double*)uu->P(sgo);
        e.seg_nseg = Cout / 16;
      }
      return e;
    };
    const std::string shape = " M" + std::to_string(M) + " Cin" + std::to_string(C) + " Cout" + std::to_string(Cout);
    const int64_t m = M * Cout * K;
    emit([=](hipStream_t s) {
      const X3Epi e = epi_of();
      return launch_gemm_bf16x3(uu->P(xo), W3, uu->P(yo), 1, (int)M, Cout, K, uu->x3_ws, s, true, false, &e);
    }, "conv k2 x3" + shape, m);
    u->macs += m;
    u->op_mfma.back() = 6 * m;   // bf16 MACs
    u->mfma_bf16_macs += 6 * m;
    if (gemm_bf16x3_needs_sum(1, (int)M, Cout, K))
      emit([=](hipStream_t s) {
        const X3Epi e = epi_of();
        return launch_gemm_bf16x3_sum(uu->P(yo), 1, (int)M, Cout, K, uu->x3_ws, s, &e);
      }, "conv k2 x3 sum" + shape);
    return y;
  }
  T downsample(const T& x, const std::string& pre, int dim_out) {  // pixel-unshuffle + conv1x1 == 2x2/s2 conv
    if (cfg.downsample_conv4) {   // earlier library versions: Conv2d(dim, dim_out, 4, stride 2, pad 1)
      ConvOpt o4;
      o4.want_seg = true;
      return conv(x, pack_conv(pre + ".weight", dim_out, x.C, x.C, 4), P(pre + ".bias", dim_out), dim_out, 4, 2, 1, o4);
    }
    const float* src = raw(pre + ".1.weight", (int64_t)dim_out * 4 * x.C);
    const int C = x.C;
    float* w = cached("unshuffle:" + pre, (size_t)dim_out * 4 * C,
                      [&](float* dst) { KD_THROW_IF(launch_pack_unshuffle(src, dst, dim_out, C, 0)); });
    if (downsample_x3_ok(x, dim_out)) return downsample_x3(x, pre, w, P(pre + ".1.bias", dim_out), dim_out);
    ConvOpt o;
    o.want_seg = true;   // feeds the GroupNorm of the level's first ResnetBlock
    return conv(x, w, P(pre + ".1.bias", dim_out), dim_out, 2, 2, 0, o);
  }
  T upsample(const T& x, const std::string& pre, int dim_out, const T* ct = nullptr) {  // conv1x1 -> SiLU -> PixelShuffle(2)
    const float* wsrc = raw(pre + ".net.0.weight", (int64_t)4 * dim_out * x.C);
    const float* bsrc = raw(pre + ".net.0.bias", 4 * dim_out);
    const int C = x.C;
    float* b = nullptr;  // weight and bias are permuted together: the bias buffer is cached under its own key
    float* w = cached("shuffle_w:" + pre, (size_t)4 * dim_out * C, [&](float* dst) {
      b = cached("shuffle_b:" + pre, (size_t)4 * dim_out, [](float*) {});
      KD_THROW_IF(launch_pack_shuffle(wsrc, bsrc, dst, b, dim_out, C, 0));
    });
    if (!b) b = cached("shuffle_b:" + pre, (size_t)4 * dim_out, [](float*) {});
    ConvOpt o;
    o.act = ACT_SILU;
    o.out_mode = OUT_PIXSHUF;
    o.want_seg = true;
    if (ct) o.dst = ct;   // straight into the first dim_out channels of the following skip concat
    return conv(x, w, b, 4 * dim_out, 1, 1, 0, o);
  }
  T concat_skip(const T& x, const T& skip, float scale) {
    T y = alloc(x.B, x.H, x.W, x.C + skip.C);
    size_t ao = x.at(), bo = skip.at(), yo = y.off;
    int Ca = x.C, Cb = skip.C, lda = x.LD(), ldb = skip.LD();
    int64_t rows = x.rows();
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      if (launch_copy_scale_rows(uu->P(ao), lda, uu->P(yo), Ca + Cb, Ca, 1.0f, rows, s)) return 1;
      return launch_copy_scale_rows(uu->P(bo), ldb, uu->P(yo) + Ca, Ca + Cb, Cb, scale, rows, s);
    }, "concat rows" + std::to_string(rows) + " C" + std::to_string(Ca + Cb));
    return y;
  }
  // the same when x's producer already wrote its Ca channels into y (resnet / upsample with ct = &y)
  void concat_skip_tail(const T& y, int Ca, const T& skip, float scale) {
    size_t bo = skip.at(), yo = y.off;
    int Cb = skip.C, ldb = skip.LD(), ldy = y.LD();
    int64_t rows = y.rows();
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      return launch_copy_scale_rows(uu->P(bo), ldb, uu->P(yo) + Ca, ldy, Cb, scale, rows, s);
    }, "concat tail rows" + std::to_string(rows) + " C" + std::to_string(Ca + Cb));
  }
  // x into the first x.C channels of the concat buffer ct (whose skip half is already there)
  void concat_head(const T& ct, const T& x) {
    size_t ao = x.at(), yo = ct.off;
    int Ca = x.C, lda = x.LD(), ldy = ct.LD();
    int64_t rows = x.rows();
    kd_unet* uu = u;
    emit([=](hipStream_t s) { return launch_copy_scale_rows(uu->P(ao), lda, uu->P(yo), ldy, Ca, 1.0f, rows, s); },
         "concat head rows" + std::to_string(rows) + " C" + std::to_string(Ca));
  }
  // in place: channels [c0, c0 + n) of t times scale (a skip half whose consumer cannot fold the scale)
  void scale_slice(const T& t, int c0, int n, float scale) {
    size_t o = t.off;
    int ld = t.LD();
    int64_t rows = t.rows();
    kd_unet* uu = u;
    emit([=](hipStream_t s) {
      return launch_copy_scale_rows(uu->P(o) + c0, ld, uu->P(o) + c0, ld, n, scale, rows, s);
    }, "scale slice rows" + std::to_string(rows) + " C" + std::to_string(n));
  }

  void collect_time_mlps() {
    std::vector<std::string> names;
    const std::string suffix = ".time_mlp.1.weight";
    for (auto& kv : params) {
      const std::string& n = kv.first;
      if (n.size() > suffix.size() && n.compare(n.size() - suffix.size(), suffix.size(), suffix) == 0)
        names.push_back(n.substr(0, n.size() - suffix.size()));
    }
    std::sort(names.begin(), names.end());
    int tcd = u->time_cond_dim;
    int total = 0;
    for (auto& pre : names) {
      int64_t ne = numel(pre + suffix);
      if (ne % tcd) throw std::runtime_error("time_mlp weight of unexpected shape: " + pre);
      tmlp_off[pre] = total;
      total += (int)(ne / tcd);
    }
    tmlp_total = total;
    tmlp_prefixes = names;
    if (total == 0) return;
    tmlp_w = cached("time_mlps_w", (size_t)total * tcd, [&](float* dst) {
      for (auto& pre : names) {
        int64_t ne = numel(pre + suffix);
        KD_HIP_THROW(hipMemcpyAsync(dst + (size_t)tmlp_off[pre] * tcd, raw(pre + suffix), (size_t)ne * 4,
                                    hipMemcpyDeviceToDevice, 0));
      }
    });
    tmlp_b = cached("time_mlps_b", (size_t)total, [&](float* dst) {
      for (auto& pre : names) {
        int64_t ne = numel(pre + suffix);
        KD_HIP_THROW(hipMemcpyAsync(dst + tmlp_off[pre], raw(pre + ".time_mlp.1.bias", ne / tcd),
                                    (size_t)(ne / tcd) * 4, hipMemcpyDeviceToDevice, 0));
      }
    });
  }

  // one time-conditioning trio: log_snr[B] -> (t [B,tcd] written/accumulated, tokens into c rows)
  void time_trio(const std::string& hid, const std::string& cond, const std::string& tok, bool lowres,
                 const T& t_out, const T& c_tok, int tok_row, const T& emb, const T& hidden) {
    int tcd = u->time_cond_dim, half = cfg.sinu_dim / 2, sw = cfg.sinu_dim + 1;
    int cd = cfg.cond_dim, ntt = cfg.num_time_tokens, ntok = c_tok.H * c_tok.W;
    const float* sw_w = P(hid + ".0.weights", half);
    kd_unet* uu = u;
    size_t eo = emb.off;
    int Bx = B;
    emit([=](hipStream_t s) {
      const float* t = lowres ? uu->in_lowres_log_snr : uu->in_log_snr;
      if (!t) {
        set_error(lowres ? "lowres_log_snr is required by this UNet" : "log_snr is required");
        return 1;
      }
      return launch_sinu_emb(t, sw_w, uu->P(eo), Bx, half, s);
    });
    skinny(emb.off, sw, P(hid + ".1.weight", (int64_t)tcd * sw), P(hid + ".1.bias", tcd), hidden.off, tcd, B, sw,
           tcd, ACT_NONE, ACT_SILU);
    // tokens: [B, ntt*cd] written at row tok_row of c (row stride ntok*cd per batch)
    skinny(hidden.off, tcd, P(tok + ".0.weight", (int64_t)cd * ntt * tcd), P(tok + ".0.bias", cd * ntt),
           c_tok.off + (size_t)tok_row * cd * sizeof(float), ntok * cd, B, tcd, cd * ntt, ACT_NONE, ACT_NONE);
    skinny(hidden.off, tcd, P(cond + ".0.weight", (int64_t)tcd * tcd), P(cond + ".0.bias", tcd), t_out.off, tcd, B,
           tcd, tcd, ACT_NONE, ACT_NONE);
  }

  void build();
  void build_text();
};

}  // namespace kd

#include "unet_build.inc"  // Builder::build(): the walk over the module tree
#include "text_build.inc"  // Builder::build_text(): step-invariant text conditioning
#include "api.inc"         // sampler loop + extern "C" entry points

