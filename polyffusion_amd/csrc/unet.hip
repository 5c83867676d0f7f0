// unet.hip - host-side plan of the denoiser forward and the pf_unet_* C ABI.
//
// The architecture walk follows the reference constructor (unet.py:30-149) and forward
// (unet.py:171-196); every arithmetic step is a launch of one of the gfx950 kernels in
// conv_mfma.hip / attention.hip / norm_stats.hip / small_kernels.hip.  No device memory is
// allocated here: weights live in ONE caller-owned packed blob, activations in a
// caller-owned workspace carved by two bump allocators (persistent block outputs = the
// U-Net skips; per-layer temporaries that are recycled layer after layer).
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <memory>
#include <vector>
#include "pf_internal.h"

namespace pf {

static thread_local std::string g_err;
int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

enum DestKind { D_RAW = 0, D_GEMM = 1, D_GEGLU_W = 2, D_GEGLU_B = 3, D_CONVOUT = 4, D_UPFOLD = 5, D_WINO = 6 };
struct Dest { int kind; size_t off; int taps, K, N, Npad, n_off; };
struct ParamSpec { std::string key; std::vector<int64_t> shape; std::vector<Dest> dests; bool packed = false; };

struct Layer {
  int kind;  // 0 conv_in, 1 res, 2 st, 3 down, 4 up
  int cin, cout;
  int hw_h = 0, hw_w = 0;                                             // res: spatial size of the level (decides whether the Winograd packings exist)
  // offsets (floats) into the packed blob
  size_t gn1_g, gn1_b, w1, b1, gn2_g, gn2_b, w2, b2, wskip, bskip;  // res; conv: w1/b1
  size_t wfold = 0;                                                   // upsample: parity-folded bf16x3 packing (16 taps)
  size_t wino1 = 0, wino2 = 0;                                        // res: Winograd packings of the two 3x3 convs (0: none)
  int emb_off;                                                        // column offset into the all-ResBlock time-bias matrix
  // spatial transformer
  size_t norm_g, norm_b, pin_w, pin_b, pout_w, pout_b;
  struct TB { int cross_off; size_t n1g, n1b, n2g, n2b, n3g, n3b, qkv, o1w, o1b, q2, kv2, o2w, o2b, v2raw, o2raw, ff1w, ff1b, ff2w, ff2b; };
  std::vector<TB> tbs;
  int st_index;
};
struct Block { std::vector<Layer> layers; };

}  // namespace pf

using namespace pf;

struct pf_unet {
  pf_unet_cfg cfg;
  std::vector<Block> in_blocks, out_blocks;
  Block mid;
  std::vector<int> skip_ch;
  int final_ch = 0, n_st = 0, sum_emb = 0, d_t = 0;
  std::vector<ParamSpec> params;
  std::map<std::string, int> index;
  size_t blob_floats = 0;
  size_t te_w0, te_b0, te_w2, te_b2, emb_w, emb_b, out_g, out_b, out_w, out_bias, in_w, in_b;
  // n_cond == 1 cross-attention collapse: to_v / to_out / bias of ALL transformer blocks, contiguous
  size_t cross_v = 0, cross_o = 0, cross_b = 0;
  int cross_total = 0;          // sum of C over transformer blocks
  bool cross_uniform = true;    // every block has the same C -> two grouped launches
  int cross_c = 0;
  int cross_cursor = 0;
  size_t cross_o_cursor = 0;
  const float* wdev = nullptr;
  void* amax_slot = nullptr;    // pf_unet_track_absmax: caller-owned device word, nullptr = off
  int opt[PF_OPT_COUNT] = {PF_OPT_AUTO, PF_OPT_AUTO, PF_OPT_AUTO, PF_OPT_AUTO, PF_OPT_AUTO};   // pf_unet_set_option
  // profiling
  int precision = PF_PREC_F32;
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> pkind;
  std::vector<double> pflops, pdirect;   // per launch: operations executed / operations of the direct form (differ for Winograd launches)
  int n_prof = 0;

  size_t alloc(size_t nfloats) { size_t o = blob_floats; blob_floats += (nfloats + 63) / 64 * 64; return o; }
  ParamSpec& add(const std::string& key, std::vector<int64_t> shape) {
    index[key] = (int)params.size();
    params.push_back(ParamSpec{key, shape, {}, false});
    return params.back();
  }
  size_t add_raw(const std::string& key, std::vector<int64_t> shape) {
    size_t n = 1; for (auto s : shape) n *= (size_t)s;
    size_t off = alloc(n);
    add(key, shape).dests.push_back(Dest{D_RAW, off, 1, 0, 0, 0, 0});
    return off;
  }
  // raw copy into a pre-allocated region at a float offset
  void add_raw_at(const std::string& key, std::vector<int64_t> shape, size_t off) {
    add(key, shape).dests.push_back(Dest{D_RAW, off, 1, 0, 0, 0, 0});
  }
  static size_t gemm_floats(int taps, int K, int N) { return (size_t)taps * K * ((N + 63) / 64 * 64); }
  // every GEMM weight is stored twice, back to back: fp32 packing, then the bf16x3 packing (same byte count)
  static size_t gemm_alloc(int taps, int K, int N) { return 2 * gemm_floats(taps, K, N); }
  size_t add_gemm(const std::string& key, int N, int K, int taps) {
    size_t off = alloc(gemm_alloc(taps, K, N));
    std::vector<int64_t> shape = taps == 1 ? std::vector<int64_t>{N, K} : std::vector<int64_t>{N, K, 3, 3};
    add(key, shape).dests.push_back(Dest{D_GEMM, off, taps, K, N, (N + 63) / 64 * 64, 0});
    return off;
  }
};

namespace pf {

static void build_res(pf_unet* u, const std::string& p, Layer& L) {
  const int ci = L.cin, co = L.cout;
  L.gn1_g = u->add_raw(p + ".in_layers.0.weight", {ci});
  L.gn1_b = u->add_raw(p + ".in_layers.0.bias", {ci});
  L.w1 = u->add_gemm(p + ".in_layers.2.weight", co, ci, 9);
  // the Winograd F(2x2, 3x3) packing beside it where the fused form can run (conv_wino.hip: 16x16-pixel tiles, 64-channel blocks)
  const bool wino_ok = L.hw_h >= 32 && L.hw_w >= 32 && L.hw_h % 16 == 0 && L.hw_w % 16 == 0 && co % 64 == 0 && ci % 32 == 0 && ci <= 1024;
  if (wino_ok) {
    L.wino1 = u->alloc((size_t)16 * ci * co);
    u->params.back().dests.push_back(Dest{D_WINO, L.wino1, 9, ci, co, co, 0});
  }
  L.b1 = u->add_raw(p + ".in_layers.2.bias", {co});
  L.emb_off = u->sum_emb;
  u->sum_emb += co;
  L.gn2_g = u->add_raw(p + ".out_layers.0.weight", {co});
  L.gn2_b = u->add_raw(p + ".out_layers.0.bias", {co});
  L.w2 = u->add_gemm(p + ".out_layers.3.weight", co, co, 9);
  if (wino_ok && ci == co) {   // (a channel-changing block folds its 1x1 skip projection into this conv: direct form only)
    L.wino2 = u->alloc((size_t)16 * co * co);
    u->params.back().dests.push_back(Dest{D_WINO, L.wino2, 9, co, co, co, 0});
  }
  L.b2 = u->add_raw(p + ".out_layers.3.bias", {co});
  if (ci != co) {
    L.wskip = u->add_gemm(p + ".skip_connection.weight", co, ci, 1);
    u->params.back().shape = {co, ci, 1, 1};
    L.bskip = u->add_raw(p + ".skip_connection.bias", {co});
  }
}

static void build_st(pf_unet* u, const std::string& p, Layer& L) {
  const int C = L.cin, dc = u->cfg.d_cond;
  L.st_index = u->n_st++;
  L.norm_g = u->add_raw(p + ".norm.weight", {C});
  L.norm_b = u->add_raw(p + ".norm.bias", {C});
  L.pin_w = u->add_gemm(p + ".proj_in.weight", C, C, 1);
  u->params.back().shape = {C, C, 1, 1};
  L.pin_b = u->add_raw(p + ".proj_in.bias", {C});
  for (int i = 0; i < u->cfg.tf_layers; ++i) {
    Layer::TB t{};
    const std::string tb = p + ".transformer_blocks." + std::to_string(i);
    // attn1: q,k,v fused into one [C -> 3C] matrix
    t.qkv = u->alloc(pf_unet::gemm_alloc(1, C, 3 * C));
    const char* nm[3] = {".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_v.weight"};
    for (int j = 0; j < 3; ++j)
      u->add(tb + nm[j], {C, C}).dests.push_back(Dest{D_GEMM, t.qkv, 1, C, C, (3 * C + 63) / 64 * 64, j * C});
    t.o1w = u->add_gemm(tb + ".attn1.to_out.0.weight", C, C, 1);
    t.o1b = u->add_raw(tb + ".attn1.to_out.0.bias", {C});
    // attn2: general (n_cond > 1) form + collapsed (n_cond == 1) raw form of to_v / to_out
    t.q2 = u->add_gemm(tb + ".attn2.to_q.weight", C, C, 1);
    t.kv2 = u->alloc(pf_unet::gemm_alloc(1, dc, 2 * C));
    u->add(tb + ".attn2.to_k.weight", {C, dc}).dests.push_back(Dest{D_GEMM, t.kv2, 1, dc, C, (2 * C + 63) / 64 * 64, 0});
    {
      ParamSpec& ps = u->add(tb + ".attn2.to_v.weight", {C, dc});
      ps.dests.push_back(Dest{D_GEMM, t.kv2, 1, dc, C, (2 * C + 63) / 64 * 64, C});
      t.cross_off = u->cross_cursor;
      u->cross_cursor += C;
      t.v2raw = u->cross_v + (size_t)t.cross_off * dc;
      u->params[u->index[tb + ".attn2.to_v.weight"]].dests.push_back(Dest{D_RAW, t.v2raw, 1, 0, 0, 0, 0});
    }
    t.o2w = u->add_gemm(tb + ".attn2.to_out.0.weight", C, C, 1);
    t.o2raw = u->cross_o + u->cross_o_cursor;
    u->cross_o_cursor += (size_t)C * C;
    u->params.back().dests.push_back(Dest{D_RAW, t.o2raw, 1, 0, 0, 0, 0});
    t.o2b = u->add_raw(tb + ".attn2.to_out.0.bias", {C});
    u->params.back().dests.push_back(Dest{D_RAW, u->cross_b + (size_t)t.cross_off, 1, 0, 0, 0, 0});
    t.n1g = u->add_raw(tb + ".norm1.weight", {C}); t.n1b = u->add_raw(tb + ".norm1.bias", {C});
    t.n2g = u->add_raw(tb + ".norm2.weight", {C}); t.n2b = u->add_raw(tb + ".norm2.bias", {C});
    t.n3g = u->add_raw(tb + ".norm3.weight", {C}); t.n3b = u->add_raw(tb + ".norm3.bias", {C});
    t.ff1w = u->alloc(pf_unet::gemm_alloc(1, C, 8 * C));
    u->add(tb + ".ff.net.0.proj.weight", {8 * C, C}).dests.push_back(Dest{D_GEGLU_W, t.ff1w, 1, C, 8 * C, 8 * C, 0});
    t.ff1b = u->alloc((size_t)8 * C);
    u->add(tb + ".ff.net.0.proj.bias", {8 * C}).dests.push_back(Dest{D_GEGLU_B, t.ff1b, 1, 0, 8 * C, 8 * C, 0});
    t.ff2w = u->add_gemm(tb + ".ff.net.2.weight", C, 4 * C, 1);
    t.ff2b = u->add_raw(tb + ".ff.net.2.bias", {C});
    L.tbs.push_back(t);
  }
  L.pout_w = u->add_gemm(p + ".proj_out.weight", C, C, 1);
  u->params.back().shape = {C, C, 1, 1};
  L.pout_b = u->add_raw(p + ".proj_out.bias", {C});
}

static void build_layer(pf_unet* u, const std::string& p, Layer& L) {
  switch (L.kind) {
    case 0:
      u->in_w = u->add_raw(p + ".weight", {L.cout, L.cin, 3, 3});
      u->in_b = u->add_raw(p + ".bias", {L.cout});
      break;
    case 1: build_res(u, p, L); break;
    case 2: build_st(u, p, L); break;
    case 3:
      L.w1 = u->add_gemm(p + ".op.weight", L.cout, L.cin, 9);
      L.b1 = u->add_raw(p + ".op.bias", {L.cout});
      break;
    case 4:
      L.w1 = u->add_gemm(p + ".conv.weight", L.cout, L.cin, 9);
      if (L.cin % 8 == 0) {   // bf16x3 mode runs the layer as four 2x2 convs on the source grid
        L.wfold = u->alloc(pf_unet::gemm_floats(16, L.cin, L.cout));
        u->params.back().dests.push_back(Dest{D_UPFOLD, L.wfold, 16, L.cin, L.cout, (L.cout + 63) / 64 * 64, 0});
      }
      L.b1 = u->add_raw(p + ".conv.bias", {L.cout});
      break;
  }
}

static bool in_list(const int32_t* v, int n, int x) {
  for (int i = 0; i < n; ++i) if (v[i] == x) return true;
  return false;
}

static int build(pf_unet* u) {
  const pf_unet_cfg& c = u->cfg;
  PF_REQUIRE(c.n_levels >= 1 && c.n_levels <= 8 && c.n_attention_levels >= 0 && c.n_attention_levels <= 8, "unet: bad level counts");
  PF_REQUIRE(c.channels > 0 && c.channels % 32 == 0, "unet: channels must be a multiple of 32 (GroupNorm32), got %d", c.channels);
  PF_REQUIRE(c.in_channels >= 1 && c.out_channels >= 1 && c.out_channels <= 4, "unet: out_channels must be 1..4");
  PF_REQUIRE(c.n_heads > 0 && c.tf_layers >= 1 && c.d_cond > 0 && c.d_cond % 4 == 0, "unet: bad attention config");
  PF_REQUIRE(c.img_h % (1 << (c.n_levels - 1)) == 0 && c.img_w % (1 << (c.n_levels - 1)) == 0, "unet: image size must be divisible by 2^(levels-1)");
  u->d_t = c.channels * 4;
  u->te_w0 = u->add_raw("time_embed.0.weight", {u->d_t, c.channels});
  u->te_b0 = u->add_raw("time_embed.0.bias", {u->d_t});
  u->te_w2 = u->add_raw("time_embed.2.weight", {u->d_t, u->d_t});
  u->te_b2 = u->add_raw("time_embed.2.bias", {u->d_t});

  int ch = c.channels;
  std::vector<int> stack;
  std::vector<int> widths;
  for (int i = 0; i < c.n_levels; ++i) widths.push_back(c.channels * c.channel_multipliers[i]);
  {
    Block b; Layer L{}; L.kind = 0; L.cin = c.in_channels; L.cout = ch; b.layers.push_back(L);
    u->in_blocks.push_back(b); stack.push_back(ch);
  }
  for (int lvl = 0; lvl < c.n_levels; ++lvl) {
    const bool att = in_list(c.attention_levels, c.n_attention_levels, lvl);
    for (int r = 0; r < c.n_res_blocks; ++r) {
      Block b; Layer L{}; L.kind = 1; L.cin = ch; L.cout = widths[lvl]; L.hw_h = c.img_h >> lvl; L.hw_w = c.img_w >> lvl; b.layers.push_back(L);
      ch = widths[lvl];
      if (att) {
        PF_REQUIRE(ch % c.n_heads == 0 && (ch / c.n_heads == 32 || ch / c.n_heads == 64), "unet: d_head %d unsupported (32 or 64)", ch / c.n_heads);
        Layer S{}; S.kind = 2; S.cin = S.cout = ch; b.layers.push_back(S);
      }
      u->in_blocks.push_back(b); stack.push_back(ch);
    }
    if (lvl != c.n_levels - 1) {
      Block b; Layer L{}; L.kind = 3; L.cin = L.cout = ch; b.layers.push_back(L);
      u->in_blocks.push_back(b); stack.push_back(ch);
    }
  }
  {
    PF_REQUIRE(ch % c.n_heads == 0 && (ch / c.n_heads == 32 || ch / c.n_heads == 64), "unet: d_head %d unsupported (32 or 64)", ch / c.n_heads);
    Layer a{}; a.kind = 1; a.cin = a.cout = ch;
    Layer s{}; s.kind = 2; s.cin = s.cout = ch;
    Layer d{}; d.kind = 1; d.cin = d.cout = ch;
    u->mid.layers = {a, s, d};
  }
  for (int lvl = c.n_levels - 1; lvl >= 0; --lvl) {
    const bool att = in_list(c.attention_levels, c.n_attention_levels, lvl);
    for (int j = 0; j <= c.n_res_blocks; ++j) {
      const int sk = stack.back(); stack.pop_back();
      u->skip_ch.push_back(sk);
      Block b; Layer L{}; L.kind = 1; L.cin = ch + sk; L.cout = widths[lvl]; L.hw_h = c.img_h >> lvl; L.hw_w = c.img_w >> lvl; b.layers.push_back(L);
      ch = widths[lvl];
      if (att) { Layer S{}; S.kind = 2; S.cin = S.cout = ch; b.layers.push_back(S); }
      if (lvl != 0 && j == c.n_res_blocks) { Layer U{}; U.kind = 4; U.cin = U.cout = ch; b.layers.push_back(U); }
      u->out_blocks.push_back(b);
    }
  }
  u->final_ch = ch;

  // n_cond == 1 cross-attention collapse: one contiguous region for all blocks' to_v / to_out / bias
  {
    size_t o_floats = 0;
    auto visit = [&](const Layer& L) {
      if (L.kind != 2) return;
      for (int i = 0; i < c.tf_layers; ++i) {
        if (u->cross_c == 0) u->cross_c = L.cin;
        if (L.cin != u->cross_c) u->cross_uniform = false;
        u->cross_total += L.cin;
        o_floats += (size_t)L.cin * L.cin;
      }
    };
    for (auto& b : u->in_blocks) for (auto& L : b.layers) visit(L);
    for (auto& L : u->mid.layers) visit(L);
    for (auto& b : u->out_blocks) for (auto& L : b.layers) visit(L);
    u->cross_v = u->alloc((size_t)u->cross_total * c.d_cond);
    u->cross_o = u->alloc(o_floats);
    u->cross_b = u->alloc((size_t)u->cross_total);
  }

  // parameter table in the reference key order
  for (size_t bi = 0; bi < u->in_blocks.size(); ++bi)
    for (size_t li = 0; li < u->in_blocks[bi].layers.size(); ++li)
      build_layer(u, "input_blocks." + std::to_string(bi) + "." + std::to_string(li), u->in_blocks[bi].layers[li]);
  for (size_t li = 0; li < u->mid.layers.size(); ++li) build_layer(u, "middle_block." + std::to_string(li), u->mid.layers[li]);
  for (size_t bi = 0; bi < u->out_blocks.size(); ++bi)
    for (size_t li = 0; li < u->out_blocks[bi].layers.size(); ++li)
      build_layer(u, "output_blocks." + std::to_string(bi) + "." + std::to_string(li), u->out_blocks[bi].layers[li]);
  u->out_g = u->add_raw("out.0.weight", {ch});
  u->out_b = u->add_raw("out.0.bias", {ch});
  u->out_w = u->alloc((size_t)c.out_channels * 9 * ch);
  u->add("out.2.weight", {c.out_channels, ch, 3, 3}).dests.push_back(Dest{D_CONVOUT, u->out_w, 9, ch, c.out_channels, 0, 0});
  u->out_bias = u->add_raw("out.2.bias", {c.out_channels});

  // all ResBlock emb_layers concatenated into one [sum_emb][d_t] matrix (+ bias) for a single mat-vec launch
  u->emb_w = u->alloc((size_t)u->sum_emb * u->d_t);
  u->emb_b = u->alloc((size_t)u->sum_emb);
  auto add_emb = [&](const std::string& p, Layer& L) {
    if (L.kind != 1) return;
    u->add_raw_at(p + ".emb_layers.1.weight", {L.cout, u->d_t}, u->emb_w + (size_t)L.emb_off * u->d_t);
    u->add_raw_at(p + ".emb_layers.1.bias", {L.cout}, u->emb_b + L.emb_off);
  };
  for (size_t bi = 0; bi < u->in_blocks.size(); ++bi)
    for (size_t li = 0; li < u->in_blocks[bi].layers.size(); ++li)
      add_emb("input_blocks." + std::to_string(bi) + "." + std::to_string(li), u->in_blocks[bi].layers[li]);
  for (size_t li = 0; li < u->mid.layers.size(); ++li) add_emb("middle_block." + std::to_string(li), u->mid.layers[li]);
  for (size_t bi = 0; bi < u->out_blocks.size(); ++bi)
    for (size_t li = 0; li < u->out_blocks[bi].layers.size(); ++li)
      add_emb("output_blocks." + std::to_string(bi) + "." + std::to_string(li), u->out_blocks[bi].layers[li]);
  return PF_OK;
}

// ---- weight repacking (host) ----
static void pack_gemm(float* dst, const float* src, int n_src, int K, int taps, int Npad, int n_off) {
  for (int n = 0; n < n_src; ++n)
    for (int k = 0; k < K; ++k)
      for (int t = 0; t < taps; ++t)
        dst[(((size_t)t * (K / 4) + k / 4) * Npad + n_off + n) * 4 + (k & 3)] = src[((size_t)n * K + k) * taps + t];
}
static inline int geglu_col(int n, int inner) {  // torch row n of ff.net.0.proj -> packed column
  const int j = n < inner ? n : n - inner;
  return 64 * (j / 32) + (n < inner ? 0 : 32) + (j % 32);
}

// (fp16 build only: bf16 pieces have fp32's range)
#define X3_RANGE_MSG "%s: a weight exceeds what this library's fp16 split packing holds (|w| <= 255.8; weights are stored times 2^8): load the checkpoint with the default library (bf16x3 / f32)"
static int pack_one(const ParamSpec& ps, const float* src, float* blob) {
  size_t numel = 1; for (auto s : ps.shape) numel *= (size_t)s;
  bool fits = true;
  for (const Dest& d : ps.dests) {
    float* dst = blob + d.off;
    switch (d.kind) {
      case D_RAW: memcpy(dst, src, numel * sizeof(float)); break;
      case D_GEMM:
        pack_gemm(dst, src, d.N, d.K, d.taps, d.Npad, d.n_off);
        if (d.K % 8 == 0) fits = pack_gemm_bf3(dst + (size_t)d.taps * d.K * d.Npad, src, d.N, d.K, d.taps, d.Npad, d.n_off, nullptr) && fits;
        break;
      case D_UPFOLD: fits = pack_upfold_bf3(dst, src, d.N, d.K, d.Npad) && fits; break;
      case D_WINO: fits = pack_wino_bf3(dst, src, d.N, d.K) && fits; break;
      case D_GEGLU_W: {
        const int inner = d.N / 2;
        for (int n = 0; n < d.N; ++n)
          for (int k = 0; k < d.K; ++k) dst[((size_t)(k / 4) * d.Npad + geglu_col(n, inner)) * 4 + (k & 3)] = src[(size_t)n * d.K + k];
        std::vector<int> cm(d.N);
        for (int n = 0; n < d.N; ++n) cm[n] = geglu_col(n, inner);
        fits = pack_gemm_bf3(dst + (size_t)d.K * d.Npad, src, d.N, d.K, 1, d.Npad, 0, cm.data()) && fits;
        break;
      }
      case D_GEGLU_B: {
        const int inner = d.N / 2;
        for (int n = 0; n < d.N; ++n) dst[geglu_col(n, inner)] = src[n];
        break;
      }
      case D_CONVOUT:  // [Cout][Cin][3][3] -> [9][Cin][Cout]
        for (int co = 0; co < d.N; ++co)
          for (int ci = 0; ci < d.K; ++ci)
            for (int t = 0; t < 9; ++t) dst[((size_t)t * d.K + ci) * d.N + co] = src[((size_t)co * d.K + ci) * 9 + t];
        break;
    }
  }
  return fits ? PF_OK : PF_EINVAL;
}

// ---- forward ----
// A tensor in the workspace: NHWC data + (optionally) the per-tile channel statistics its producer emitted.
struct Tn {
  const float* d = nullptr; int c = 0;
  const float* st = nullptr; int nt = 0;   // [B][nt][c][2] (sum, sumsq) or null
  int bmod = 0;                            // > 0: the tensor (and its statistics) holds only bmod samples, shared by samples b and b + bmod (pf_unet_forward_cfg)
};

struct Ctx {
  pf_unet* u; hipStream_t s; bool dry;
  char* base; size_t persist_off, temp_base, temp_off, persist_max, temp_max;
  int B, n_cond;
  const float* W;
  int n_launch;
  int rc;
  // hoisted step-invariant prefix (pf_unet_prepared): supplied parts are not recomputed; dry runs only need to know WHETHER they are
  bool has_time = false, has_cross = false;
  const float* prep_time = nullptr; int prep_time_rows = 0; const float* prep_cross = nullptr;
  const int64_t* t_rows = nullptr;   // with a time table: the per-sample row index = t
  // classifier-free guidance with a shared prefix (pf_unet_forward_cfg): while `shared` the plan runs on the first Bfull / 2 samples only
  bool cfg_share = false, shared = false; int Bfull = 0;
  int x1mod(const Tn& x1) const { return (x1.c > 0 && x1.bmod > 0 && x1.bmod != B) ? x1.bmod : 0; }

  float* palloc(size_t nfloats) {
    size_t o = persist_off; persist_off += align_up(nfloats * 4, 256);
    if (persist_off > persist_max) persist_max = persist_off;
    return dry ? nullptr : (float*)(base + o);
  }
  float* talloc(size_t nfloats) {
    size_t o = temp_off; temp_off += align_up(nfloats * 4, 256);
    if (temp_off > temp_max) temp_max = temp_off;
    return dry ? nullptr : (float*)(base + temp_base + o);
  }
  void treset() { temp_off = 0; }
  const float* w(size_t off) const { return dry ? nullptr : W + off; }

  void prof_begin(int kind, double flops, double direct = -1.0) {
    ++n_launch;
    if (dry || !u->profiling) return;
    u->pdirect.push_back(direct < 0.0 ? flops : direct);
    const size_t need = (size_t)(u->n_prof + 1) * 2;
    while (u->ev.size() < need) { hipEvent_t e; (void)hipEventCreate(&e); u->ev.push_back(e); }
    u->pkind.push_back(kind); u->pflops.push_back(flops);
    (void)hipEventRecord(u->ev[(size_t)u->n_prof * 2], s);
  }
  void prof_end() {
    if (dry || !u->profiling) return;
    (void)hipEventRecord(u->ev[(size_t)u->n_prof * 2 + 1], s);
    ++u->n_prof;
  }
  // launch a conv/linear; when `stats` is given, the producer also emits per-tile channel statistics for a later GroupNorm
  // (buffer from the persistent or the temp region, matching the lifetime of the output tensor)
  void conv(pf_conv_args a, int kind, Tn* stats = nullptr, bool persist = true, const float* w_bf3 = nullptr) {
    const int cin_ = a.c0 + a.c1;
    const bool bf3 = u->precision == PF_PREC_BF16X3 && cin_ % 32 == 0;
    if (bf3) a.precision = PF_PREC_BF16X3;   // decided before the tile (and thus the statistics layout) is chosen
    a.absmax_slot = u->amax_slot;
    a.no_t16 = u->opt[PF_OPT_CONV_T16] == PF_OPT_OFF;
    a.no_pp = u->opt[PF_OPT_CONV_PP] == PF_OPT_OFF;
    if (const size_t wsb = conv_splitk_ws_bytes(a)) {   // small-M layer: K-split partial sums live in the temp region
      float* ws = talloc(wsb / 4);
      a.splitk_ws = dry ? (void*)1 : (void*)ws; a.splitk_ws_bytes = wsb;
    }
    if (stats) {
      const int nt = conv_stats_tiles(a);
      float* sb = persist ? palloc((size_t)B * nt * a.n * 2) : talloc((size_t)B * nt * a.n * 2);
      a.stats_out = sb;
      stats->d = a.out; stats->c = a.n; stats->st = sb; stats->nt = nt;
    }
    double direct = -1.0;
    if (a.wino) { pf_conv_args d = a; d.wino = 0; direct = conv_flops(d); }
    prof_begin(kind, conv_flops(a), direct);
    if (!dry && rc == PF_OK) {
      if (w_bf3) a.w = w_bf3;                                                       // a packing of its own (folded upsampling conv)
      else if (bf3) a.w = a.w + (size_t)a.ks * a.ks * cin_ * ((a.n + 63) / 64 * 64);  // second half of the region = bf16x3 packing
      rc = launch_conv(a, s);
    }
    prof_end();
  }
  // GroupNorm scale/shift of concat(x0, x1) from the producers' tile statistics (no pass over the data).
  // `fuse_ok`: the consumer is a bf16x3 conv that can do this reduction in its own prologue (pf_conv_args.gn_*): worth it when a
  // sample has few statistics tiles (the 32x32 / 16x16 levels: <= 16 tiles), where the 5 us finalize launch is 10-20 % of the
  // convolution it feeds; at the 128x128 / 64x64 levels every consumer workgroup would re-read 32-64 KB, so the launch stays.
  struct GnRef { bool fused = false; const float* s0 = nullptr; const float* s1 = nullptr; int t0 = 0, t1 = 0; size_t g = 0, b = 0; float eps = 0.f; int bmod1 = 0; };
  GnRef gn(const Tn& x0, const Tn& x1, int hw, float eps, size_t g, size_t b_, float* sc, float* sh, bool fuse_ok = false) {
    const int cin_ = x0.c + x1.c;
    // (<= 32 tiles - the 64x64 level too - measured neutral in round 5 with the batched statistics loads, -0.9 % before them; <= 128: -3 %)
    constexpr int fold_max = 16;
    if (fuse_ok && u->precision == PF_PREC_BF16X3 && cin_ % 32 == 0 && cin_ <= 1024 && x0.nt <= fold_max && (x1.c == 0 || x1.nt <= fold_max)) {
      GnRef r; r.fused = true; r.s0 = x0.st; r.t0 = x0.nt; r.s1 = x1.st; r.t1 = x1.nt; r.g = g; r.b = b_; r.eps = eps; r.bmod1 = x1mod(x1);
      return r;
    }
    gn_launch(x0, x1, hw, eps, g, b_, sc, sh);
    return GnRef{};
  }
  // the fused Winograd form of a ResBlock conv (PF_OPT_CONV_WINO).  AUTO follows the same-box A/B of profiles/r06_ab_winograd.md: the form
  // wins where the K loop is long enough to carry its per-tile exchange - 192 input channels and more, or 128 and more from the 32x32 level
  // down - and when its 16x16-pixel x 64-channel workgroups fill at least three quarters of the CUs.
  void wino_attach(pf_conv_args& a, size_t wino_off) {
    const int o = u->opt[PF_OPT_CONV_WINO];
    if (!wino_off || o == PF_OPT_OFF || u->precision != PF_PREC_BF16X3) return;
    const int cin_ = a.c0 + a.c1;
    if (o == PF_OPT_AUTO) {
      const int wgs = a.batch * (a.hin / 16) * (a.win / 16) * (a.n / 64);
      const bool deep = cin_ >= 192 || (cin_ >= 128 && a.hin * a.win <= 1024);
      if (!deep || wgs * 4 < num_cus() * 3) return;
    }
    a.w_wino = dry ? (const void*)16 : (const void*)w(wino_off);
    a.wino = 1;
  }
  void gn_attach(pf_conv_args& a, const GnRef& r) {
    if (!r.fused) return;
    a.gn_stats0 = dry ? (const float*)16 : r.s0; a.gn_tiles0 = r.t0; a.gn_stats1 = r.s1; a.gn_tiles1 = r.t1;
    a.gn_gamma = w(r.g); a.gn_beta = w(r.b); a.gn_eps = r.eps; a.gn_groups = 32;
  }
  void gn_launch(const Tn& x0, const Tn& x1, int hw, float eps, size_t g, size_t b_, float* sc, float* sh) {
    prof_begin(PF_K_GNSTAT, 0.0);
    if (!dry && rc == PF_OK)
      rc = launch_gn_finalize_tiles(x0.st, x0.nt, x0.c, x1.st, x1.nt, x1.c, B, hw, 32, eps, w(g), w(b_), sc, sh, s, x1mod(x1));
    prof_end();
  }
  // statistics for a tensor whose producer emitted none (the stem conv output)
  void gn_partial(Tn& x, int hw) {
    const int ns = gn_nsplit(hw);
    float* sb = palloc((size_t)B * ns * x.c * 2);
    prof_begin(PF_K_GNSTAT, 0.0);
    if (!dry && rc == PF_OK) rc = launch_gn_partial(x.d, x.c, nullptr, 0, B, hw, sb, s);
    prof_end();
    x.st = sb; x.nt = ns;
  }
  void lnp(const float* x, int rows, int c, size_t gamma, size_t beta, float* planes) {   // LayerNorm -> hi/lo planes
    prof_begin(PF_K_LNSTAT, 0.0);
    if (!dry && rc == PF_OK) rc = launch_ln_planes(x, rows, c, 1e-5f, w(gamma), w(beta), planes, s);
    prof_end();
  }
  void ln(const float* x, int rows, int c, float* mu, float* rs) {
    prof_begin(PF_K_LNSTAT, 0.0);
    if (!dry && rc == PF_OK) rc = launch_ln_stats(x, rows, c, 1e-5f, mu, rs, s);
    prof_end();
  }
};

static pf_conv_args conv_base(const float* x0, int c0, const float* x1, int c1, int B, int hin, int win, int ks,
                              const float* wgt, int n, float* out) {
  pf_conv_args a;
  memset(&a, 0, sizeof a);
  a.x0 = x0; a.c0 = c0; a.x1 = x1; a.c1 = c1; a.batch = B; a.hin = hin; a.win = win; a.ks = ks; a.stride = 1;
  a.w = wgt; a.n = n; a.out = out; a.ld_out = n;
  return a;
}

static Tn run_res(Ctx& c, const Layer& L, const Tn& x0, const Tn& x1, int H, int W_, const float* tb_all) {
  const int B = c.B, hw = H * W_, ci = L.cin, co = L.cout;
  float* out = c.palloc((size_t)B * hw * co);
  c.treset();
  float* sc1 = c.talloc((size_t)B * ci); float* sh1 = c.talloc((size_t)B * ci);
  float* h = c.talloc((size_t)B * hw * co);
  float* sc2 = c.talloc((size_t)B * co); float* sh2 = c.talloc((size_t)B * co);
  const Ctx::GnRef g1 = c.gn(x0, x1, hw, 1e-5f, L.gn1_g, L.gn1_b, sc1, sh1, true);
  Tn ht;
  {
    pf_conv_args a = conv_base(x0.d, x0.c, x1.d, x1.c, B, H, W_, 3, c.w(L.w1), co, h);
    a.prologue = 1; a.sc = sc1; a.sh = sh1; a.bias = c.w(L.b1);
    c.gn_attach(a, g1);
    a.sbias = c.dry ? nullptr : tb_all + L.emb_off; a.ld_sbias = c.u->sum_emb;
    a.sbias_rows = c.t_rows; a.sbias_nrows = c.prep_time_rows;   // hoisted table: row = t[b]
    a.x1_bmod = c.x1mod(x1);
    c.wino_attach(a, L.wino1);
    c.conv(a, PF_K_CONV3, &ht, false);
  }
  const Ctx::GnRef g2 = c.gn(ht, Tn{}, hw, 1e-5f, L.gn2_g, L.gn2_b, sc2, sh2, true);
  const float* res = x0.d;
  // bf16x3: the 1x1 skip_connection conv is folded into the second 3x3 conv as one more K range (no round trip of the
  // projected tensor through HBM, one launch less)
  const bool fuse_skip = ci != co && c.u->precision == PF_PREC_BF16X3 && x0.c % 32 == 0 && x1.c % 32 == 0 && co % 32 == 0;
  if (ci != co && !fuse_skip) {
    float* sk = c.talloc((size_t)B * hw * co);
    pf_conv_args a = conv_base(x0.d, x0.c, x1.d, x1.c, B, 1, hw, 1, c.w(L.wskip), co, sk);
    a.bias = c.w(L.bskip);
    a.x1_bmod = c.x1mod(x1);
    c.conv(a, PF_K_GEMM);
    res = sk;
  }
  Tn ot;
  {
    pf_conv_args a = conv_base(h, co, nullptr, 0, B, H, W_, 3, c.w(L.w2), co, out);
    a.prologue = 1; a.sc = sc2; a.sh = sh2; a.bias = c.w(L.b2);
    c.gn_attach(a, g2);
    if (fuse_skip) {
      a.skip_x0 = x0.d; a.skip_c0 = x0.c; a.skip_x1 = x1.d; a.skip_c1 = x1.c;
      a.skip_w = c.dry ? (const void*)1 : (const void*)(c.w(L.wskip) + (size_t)ci * ((co + 63) / 64 * 64));   // its bf16x3 packing
      a.skip_bias = c.w(L.bskip);
      a.x1_bmod = c.x1mod(x1);
    } else {
      a.res = res; a.ld_res = co;
    }
    c.wino_attach(a, L.wino2);
    c.conv(a, PF_K_CONV3, &ot, true);
  }
  return ot;
}

// The fused feed-forward launch (mlp_fused_bf3.hip) gives every 64-row tile to one four-wave workgroup that streams all 3 MB of ff
// weights and occupies a whole CU (152 KB of LDS), so it runs in rounds of 256 tiles: measured 95 us per round against 113 us per
// 16384 rows for the three launches it replaces (B = 16, L = 1024).  It is used when the last round is at least 85 % full
// (B = 16 and B = 32 at the 32x32 level; never at the 16x16 level below B = 55).  pf_unet_set_option(PF_OPT_MLP_FUSED) forces it off / on.
static bool mlp_fused_wanted(int C, int hw, int M, int force) {
  if (C != 256 || hw % 64 != 0) return false;
  if (force != PF_OPT_AUTO) return force != PF_OPT_OFF;
  const int cus = num_cus(), tiles = M / 64, rounds = (tiles + cus - 1) / cus;
  return tiles * 100 >= 85 * cus * rounds;
}

static Tn run_st(Ctx& c, const Layer& L, const Tn& xin, int H, int W_, const float* cond, const float* cross_all) {
  const int B = c.B, hw = H * W_, C = L.cin, M = B * hw, nh = c.u->cfg.n_heads, dh = C / nh, dc = c.u->cfg.d_cond;
  const float* x = xin.d;
  float* out = c.palloc((size_t)M * C);
  c.treset();
  float* sc = c.talloc((size_t)B * C); float* sh = c.talloc((size_t)B * C);
  float* ta = c.talloc((size_t)M * C); float* tbuf = c.talloc((size_t)M * C); float* tc = c.talloc((size_t)M * C);
  float* mu = c.talloc(M); float* rs = c.talloc(M);
  float* qkv = c.talloc((size_t)M * 3 * C);
  float* att = c.talloc((size_t)M * C);
  float* ff = c.talloc((size_t)M * 4 * C);
  float* kv = nullptr;
  if (c.n_cond > 1) kv = c.talloc((size_t)B * c.n_cond * 2 * C);
  const Ctx::GnRef gin = c.gn(xin, Tn{}, hw, 1e-6f, L.norm_g, L.norm_b, sc, sh, true);
  // bf16x3 mode, d_head 64, L % 128 == 0: every linear layer of the block runs on pre-split hi/lo planes (see the loop below)
  const bool planes_ok = c.u->precision == PF_PREC_BF16X3 && dh == 64 && hw % 128 == 0 && C % 32 == 0 && C <= 1024;
  {
    pf_conv_args a = conv_base(x, C, nullptr, 0, B, 1, hw, 1, c.w(L.pin_w), C, ta);
    a.prologue = 2; a.sc = sc; a.sh = sh; a.bias = c.w(L.pin_b);
    c.gn_attach(a, gin);
    c.conv(a, PF_K_GEMM);
  }
  float* t0 = ta; float* t1 = tbuf; float* t2 = tc;
  bool last_planes = false;
  for (size_t i = 0; i < L.tbs.size(); ++i) {
    const Layer::TB& t = L.tbs[i];
    // x = attn1(LN1(x)) + x
    // bf16x3 mode, d_head 64, L % 128 == 0: every linear layer of the block runs as a planes GEMM (both operands stream
    // global->LDS): LayerNorm is applied once per row into hi/lo planes instead of once per column tile in a GEMM prologue,
    // the projection writes pre-split q/k/v^T planes and attention runs on the bf16 pipe
    const bool planes = planes_ok;
    if (planes) {
      c.lnp(t0, M, C, t.n1g, t.n1b, att);   // `att` is free until attention writes it
      pf_conv_args a = conv_base(att, C, nullptr, 0, B, 1, hw, 1, c.w(t.qkv), 3 * C, qkv);
      a.a_planes = 1;
      a.qkv_planes = c.dry ? (void*)1 : (void*)qkv;   // same bytes as the fp32 [M][3C] buffer
      c.conv(a, PF_K_GEMM);
    } else {
      c.ln(t0, M, C, mu, rs);
      pf_conv_args a = conv_base(t0, C, nullptr, 0, B, 1, hw, 1, c.w(t.qkv), 3 * C, qkv);
      a.prologue = 3; a.sc = c.w(t.n1g); a.sh = c.w(t.n1b); a.mean = mu; a.rstd = rs;
      c.conv(a, PF_K_GEMM);
    }
    // small batches (few query tiles, many key tiles each): key slices across workgroups + a merging launch, partial results in the temp region
    int att_ns = 1;
    const size_t att_sf = planes ? attention_bf3_split_floats(B, nh, hw, &att_ns) : 0;
    float* att_scratch = att_sf ? c.talloc(att_sf) : nullptr;
    c.prof_begin(PF_K_ATTN, 4.0 * B * nh * (double)hw * hw * dh);
    if (att_sf && c.u->opt[PF_OPT_ATTN_WIDE] != PF_OPT_ON) ++c.n_launch;   // the merge
    if (!c.dry && c.rc == PF_OK)
      c.rc = planes ? launch_attention_bf3(qkv, nullptr, C, att, B, nh, hw, c.u->opt[PF_OPT_ATTN_WIDE], c.s, att_scratch, att_sf)   // att as hi/lo planes for the to_out GEMM
                    : launch_attention(qkv, 3 * C, qkv + C, 3 * C, qkv + 2 * C, 3 * C, att, C, B, nh, dh, hw, hw, c.s);
    c.prof_end();
    last_planes = planes && (i + 1 == L.tbs.size());
    const bool fuse_mlp = planes && mlp_fused_wanted(C, hw, M, c.u->opt[PF_OPT_MLP_FUSED]) && !c.u->amax_slot;
    {
      pf_conv_args a = conv_base(att, C, nullptr, 0, B, 1, hw, 1, c.w(t.o1w), C, t1);
      a.bias = c.w(t.o1b); a.res = t0; a.ld_res = C; a.a_planes = planes ? 1 : 0;
      if (c.n_cond == 1) {  // x = attn2(LN2(x), c) + x collapses to a per-sample bias (softmax over one key == 1)
        a.sbias = c.dry ? nullptr : cross_all + t.cross_off;
        a.ld_sbias = c.u->cross_total;
      }
      c.conv(a, PF_K_GEMM);
    }
    if (c.n_cond > 1) {
      c.ln(t1, M, C, mu, rs);
      float* q2 = qkv;  // reuse: [M][C]
      {
        pf_conv_args a = conv_base(t1, C, nullptr, 0, B, 1, hw, 1, c.w(t.q2), C, q2);
        a.prologue = 3; a.sc = c.w(t.n2g); a.sh = c.w(t.n2b); a.mean = mu; a.rstd = rs;
        c.conv(a, PF_K_GEMM);
      }
      {
        pf_conv_args a = conv_base(cond, dc, nullptr, 0, B, 1, c.n_cond, 1, c.w(t.kv2), 2 * C, kv);
        c.conv(a, PF_K_GEMM);
      }
      c.prof_begin(PF_K_ATTN, 4.0 * B * nh * (double)hw * c.n_cond * dh);
      if (!c.dry && c.rc == PF_OK) c.rc = launch_attention(q2, C, kv, 2 * C, kv + C, 2 * C, att, C, B, nh, dh, hw, c.n_cond, c.s);
      c.prof_end();
      {
        pf_conv_args a = conv_base(att, C, nullptr, 0, B, 1, hw, 1, c.w(t.o2w), C, t2);
        a.bias = c.w(t.o2b); a.res = t1; a.ld_res = C;
        c.conv(a, PF_K_GEMM);
      }
      std::swap(t1, t2);
    }
    // x = ff(LN3(x)) + x
    if (fuse_mlp) {
      // one launch: LayerNorm, GeGLU projection, output projection and residual per 64-row tile, hidden tensor kept on chip
      if (last_planes) {
        // the block's proj_out rides on the same launch: ff output + residual stay in LDS as its A operand, the result lands in `out`
        // together with the 64-row-tile statistics the next GroupNorm reads
        const int nt = hw / 64;
        float* sb = c.palloc((size_t)B * nt * C * 2);
        c.prof_begin(PF_K_GEMM, 2.0 * M * ((double)C * 8 * C + 4.0 * C * C + (double)C * C));
        if (!c.dry && c.rc == PF_OK)
          c.rc = launch_mlp_fused(t1, B, hw, c.w(t.n3g), c.w(t.n3b), 1e-5f, c.w(t.ff1w) + (size_t)C * 8 * C, c.w(t.ff1b),
                                  c.w(t.ff2w) + (size_t)4 * C * C, c.w(t.ff2b), out, nullptr, c.s, c.w(L.pout_w) + (size_t)C * C, c.w(L.pout_b), x, sb);
        c.prof_end();
        Tn ot;
        ot.d = out; ot.c = C; ot.st = sb; ot.nt = nt;
        return ot;
      }
      c.prof_begin(PF_K_GEMM, 2.0 * M * ((double)C * 8 * C + 4.0 * C * C));
      if (!c.dry && c.rc == PF_OK)
        c.rc = launch_mlp_fused(t1, B, hw, c.w(t.n3g), c.w(t.n3b), 1e-5f, c.w(t.ff1w) + (size_t)C * 8 * C, c.w(t.ff1b),
                                c.w(t.ff2w) + (size_t)4 * C * C, c.w(t.ff2b), t2, nullptr, c.s);
      c.prof_end();
      std::swap(t0, t2);
      continue;
    }
    if (planes) {
      c.lnp(t1, M, C, t.n3g, t.n3b, att);   // `att` has been consumed by to_out
      pf_conv_args a = conv_base(att, C, nullptr, 0, B, 1, hw, 1, c.w(t.ff1w), 8 * C, ff);
      a.a_planes = 1; a.bias = c.w(t.ff1b); a.geglu = 1; a.ld_out = 4 * C;
      a.out_planes = c.dry ? (void*)1 : (void*)ff;   // GeGLU product straight into hi/lo planes for the ff2 GEMM
      c.conv(a, PF_K_GEMM);
    } else {
      c.ln(t1, M, C, mu, rs);
      pf_conv_args a = conv_base(t1, C, nullptr, 0, B, 1, hw, 1, c.w(t.ff1w), 8 * C, ff);
      a.prologue = 3; a.sc = c.w(t.n3g); a.sh = c.w(t.n3b); a.mean = mu; a.rstd = rs; a.bias = c.w(t.ff1b);
      a.geglu = 1; a.ld_out = 4 * C;
      c.conv(a, PF_K_GEMM);
    }
    {
      pf_conv_args a = conv_base(ff, 4 * C, nullptr, 0, B, 1, hw, 1, c.w(t.ff2w), C, t2);
      a.bias = c.w(t.ff2b); a.res = t1; a.ld_res = C; a.a_planes = planes ? 1 : 0;
      if (last_planes) a.out_planes = c.dry ? (void*)1 : (void*)t2;   // only proj_out reads it: hand it over as planes
      c.conv(a, PF_K_GEMM);
    }
    std::swap(t0, t2);
  }
  Tn ot;
  {
    pf_conv_args a = conv_base(t0, C, nullptr, 0, B, 1, hw, 1, c.w(L.pout_w), C, out);
    a.bias = c.w(L.pout_b); a.res = x; a.ld_res = C; a.a_planes = last_planes ? 1 : 0;
    c.conv(a, PF_K_GEMM, &ot, true);
  }
  return ot;
}

static void small_launch(Ctx& c, int rc_in) { if (c.rc == PF_OK) c.rc = rc_in; }

// n_cond == 1: cross[b] = to_out(to_v(cond[b])) + bias of EVERY transformer block (softmax over one key == 1, unet_attention.py:186-212):
// two grouped mat-vec launches (one per block when the blocks differ in width).  Shared by the forward plan and pf_unet_prepare_cond.
static void cross_bias_launches(pf_unet* u, Ctx& c, const float* cond, int B, float* cross, float* vtmp) {
  const int T = u->cross_total, dc = u->cfg.d_cond;
  c.prof_begin(PF_K_SMALL, 0);
  if (!c.dry) small_launch(c, launch_matvec(cond, dc, c.w(u->cross_v), nullptr, vtmp, T, B, T, dc, c.s));
  c.prof_end();
  if (u->cross_uniform) {
    c.prof_begin(PF_K_SMALL, 0);
    if (!c.dry) small_launch(c, launch_matvec(vtmp, T, c.w(u->cross_o), c.w(u->cross_b), cross, T, B, T, u->cross_c, c.s, u->cross_c, u->cross_c));
    c.prof_end();
    return;
  }
  auto each_tb = [&](const Layer& L) {
    if (L.kind != 2) return;
    for (const Layer::TB& tb : L.tbs) {
      c.prof_begin(PF_K_SMALL, 0);
      if (!c.dry) small_launch(c, launch_matvec(vtmp + tb.cross_off, T, c.w(tb.o2raw), c.w(tb.o2b), cross + tb.cross_off, T, B, L.cin, L.cin, c.s));
      c.prof_end();
    }
  };
  for (auto& b : u->in_blocks) for (auto& L : b.layers) each_tb(L);
  for (auto& L : u->mid.layers) each_tb(L);
  for (auto& b : u->out_blocks) for (auto& L : b.layers) each_tb(L);
}

static int run(pf_unet* u, Ctx& c, const float* x, const int64_t* t, const float* cond, float* eps) {
  const pf_unet_cfg& cfg = u->cfg;
  const int B = c.B;
  int H = cfg.img_h, W_ = cfg.img_w;
  // time embedding and every ResBlock's additive time bias
  // (the workspace layout does not depend on what the caller prepared: the small buffers are carved either way)
  float* tsilu = c.palloc((size_t)B * u->d_t);
  const float* tb_all = c.palloc((size_t)B * u->sum_emb);
  if (c.has_time) {   // hoisted: row t[b] of the caller's table (pf_unet_prepare_time) instead of two launches per forward
    tb_all = c.prep_time; c.t_rows = t;
  } else {
    float* tb = const_cast<float*>(tb_all);
    c.prof_begin(PF_K_SMALL, 0); if (!c.dry) small_launch(c, launch_time_embed(t, c.w(u->te_w0), c.w(u->te_b0), c.w(u->te_w2), c.w(u->te_b2), tsilu, B, cfg.channels, u->d_t, c.s)); c.prof_end();
    c.prof_begin(PF_K_SMALL, 0); if (!c.dry) small_launch(c, launch_matvec(tsilu, u->d_t, c.w(u->emb_w), c.w(u->emb_b), tb, u->sum_emb, B, u->sum_emb, u->d_t, c.s)); c.prof_end();
  }
  const float* cross_all = nullptr;  // [B][cross_total]: to_out(to_v(c)) + bias of every transformer block
  if (c.n_cond == 1 && u->cross_total > 0) {
    const int T = u->cross_total;
    float* cross_ws = c.palloc((size_t)B * T);
    float* vtmp = c.palloc((size_t)B * T);
    cross_all = cross_ws;
    if (c.has_cross) cross_all = c.prep_cross;   // hoisted (pf_unet_prepare_cond)
    else cross_bias_launches(u, c, cond, B, cross_ws, vtmp);
  }

  std::vector<Tn> skips;
  Tn cur;
  // Classifier-free guidance with a shared prefix: until the first SpatialTransformer the plan runs on the first half of the batch only
  // (the two halves are identical there: the condition has not entered yet); right before it the running tensor and its statistics are
  // duplicated and the plan continues on the whole batch.  Skip tensors produced in the shared phase keep their half size and are read
  // with the sample index modulo the half (Tn::bmod -> pf_conv_args.x1_bmod).
  if (c.cfg_share) { c.Bfull = c.B; c.B = c.B / 2; c.shared = true; }
  auto dup_rows = [&](const float* src, size_t floats_per_half) -> const float* {
    float* dst = c.palloc(2 * floats_per_half);
    for (int h = 0; h < 2; ++h) {
      c.prof_begin(PF_K_SMALL, 0);
      if (!c.dry && c.rc == PF_OK && hipMemcpyAsync(dst + h * floats_per_half, src, floats_per_half * sizeof(float), hipMemcpyDeviceToDevice, c.s) != hipSuccess)
        c.rc = set_error(PF_EHIP, "pf_unet_forward_cfg: device copy failed");
      c.prof_end();
    }
    return dst;
  };
  auto leave_shared_phase = [&](Tn& a0, int h, int w_) {
    if (!c.shared) return;
    const int Bh = c.B;
    a0.d = dup_rows(a0.d, (size_t)Bh * h * w_ * a0.c);
    if (a0.nt > 0) a0.st = dup_rows(a0.st, (size_t)Bh * a0.nt * a0.c * 2);   // (nt, not st: a dry run carries no pointers)
    a0.bmod = 0;
    c.B = c.Bfull; c.shared = false;
  };
  auto run_layers = [&](const Block& b, Tn in0, Tn in1) {
    Tn a0 = in0, a1 = in1;
    for (const Layer& L : b.layers) {
      Tn o;
      if (L.kind == 2) leave_shared_phase(a0, H, W_);
      const int B = c.B;
      switch (L.kind) {
        case 0: {
          float* od = c.palloc((size_t)B * H * W_ * L.cout);
          // the stem conv emits the per-tile channel statistics of its output itself when it can (the usual 2 -> 64 stem);
          // otherwise a statistics pass over the output follows
          const int nst = launch_conv_in_stats_tiles(L.cin, L.cout, H, W_);
          float* sb = nst ? c.palloc((size_t)B * nst * L.cout * 2) : nullptr;
          c.prof_begin(PF_K_SMALL, 2.0 * B * H * W_ * 9.0 * L.cin * L.cout);
          if (!c.dry) small_launch(c, launch_conv_in(x, c.w(u->in_w), c.w(u->in_b), od, B, L.cin, L.cout, H, W_, c.s, sb));
          c.prof_end();
          o.d = od; o.c = L.cout;
          if (nst) { o.st = sb; o.nt = nst; } else c.gn_partial(o, H * W_);
          break;
        }
        case 1: o = run_res(c, L, a0, a1, H, W_, tb_all); break;
        case 2: o = run_st(c, L, a0, H, W_, cond, cross_all); break;
        case 3: {
          float* od = c.palloc((size_t)B * (H / 2) * (W_ / 2) * L.cout);
          pf_conv_args a = conv_base(a0.d, a0.c, nullptr, 0, B, H, W_, 3, c.w(L.w1), L.cout, od);
          a.stride = 2; a.bias = c.w(L.b1);
          c.conv(a, PF_K_CONV3, &o, true);
          H /= 2; W_ /= 2;
          break;
        }
        case 4: {
          float* od = c.palloc((size_t)B * (H * 2) * (W_ * 2) * L.cout);
          pf_conv_args a = conv_base(a0.d, a0.c, nullptr, 0, B, H, W_, 3, c.w(L.w1), L.cout, od);
          a.ups = 1; a.bias = c.w(L.b1);
          if (c.u->precision == PF_PREC_BF16X3 && L.wfold && L.cin % 32 == 0) {
            a.ups_fold = 1; a.precision = PF_PREC_BF16X3;
            c.conv(a, PF_K_CONV3, &o, true, c.dry ? nullptr : c.w(L.wfold));
          } else {
            c.conv(a, PF_K_CONV3, &o, true);
          }
          H *= 2; W_ *= 2;
          break;
        }
      }
      if (c.dry) o.c = L.cout;
      if (c.shared) o.bmod = c.B;
      a0 = o; a1 = Tn{};
    }
    cur = a0;
  };

  for (const Block& b : u->in_blocks) {
    run_layers(b, cur, Tn{});
    skips.push_back(cur);
  }
  run_layers(u->mid, cur, Tn{});
  for (const Block& b : u->out_blocks) {
    const Tn sk = skips.back();
    skips.pop_back();
    run_layers(b, cur, sk);  // channel order [x, skip] (unet.py:192)
  }
  // out: GN + SiLU + conv3x3 -> NCHW
  c.treset();
  const int Bo = c.B;   // (still the half batch only for a UNet without any transformer block: eps is then duplicated below)
  float* sc = c.talloc((size_t)Bo * cur.c); float* sh = c.talloc((size_t)Bo * cur.c);
  c.gn(cur, Tn{}, H * W_, 1e-5f, u->out_g, u->out_b, sc, sh);
  c.prof_begin(PF_K_SMALL, 2.0 * Bo * H * W_ * 9.0 * cur.c * cfg.out_channels);
  if (!c.dry) small_launch(c, launch_conv_out(cur.d, sc, sh, c.w(u->out_w), c.w(u->out_bias), eps, Bo, cur.c, cfg.out_channels, H, W_, c.s));
  c.prof_end();
  if (c.shared) {   // no layer ever looked at the condition: both halves of eps are the same image
    const size_t n = (size_t)Bo * cfg.out_channels * H * W_;
    c.prof_begin(PF_K_SMALL, 0);
    if (!c.dry && c.rc == PF_OK && hipMemcpyAsync(eps + n, eps, n * sizeof(float), hipMemcpyDeviceToDevice, c.s) != hipSuccess)
      c.rc = set_error(PF_EHIP, "pf_unet_forward_cfg: device copy failed");
    c.prof_end();
    c.B = c.Bfull; c.shared = false;
  }
  return c.rc;
}

static void plan_sizes(pf_unet* u, int batch, int n_cond, size_t* persist, size_t* temp, int* launches, bool has_time = false, bool has_cross = false,
                       bool cfg_share = false) {
  Ctx c{};
  c.u = u; c.dry = true; c.B = batch; c.n_cond = n_cond; c.rc = PF_OK; c.has_time = has_time; c.has_cross = has_cross; c.cfg_share = cfg_share;
  run(u, c, nullptr, nullptr, nullptr, nullptr);
  *persist = align_up(c.persist_max, 4096);
  *temp = align_up(c.temp_max, 4096);
  if (launches) *launches = c.n_launch;
}

}  // namespace pf

extern "C" {

int pf_version(void) { return 100; }
const char* pf_last_error(void) { return g_err.c_str(); }

int pf_unet_create(const pf_unet_cfg* cfg, pf_unet** out) {
  PF_REQUIRE(cfg && out, "pf_unet_create: null argument");
  std::unique_ptr<pf_unet> u(new pf_unet());
  u->cfg = *cfg;
  int rc = build(u.get());
  if (rc != PF_OK) return rc;
  *out = u.release();
  return PF_OK;
}

void pf_unet_destroy(pf_unet* u) {
  if (!u) return;
  for (auto e : u->ev) (void)hipEventDestroy(e);
  delete u;
}

size_t pf_unet_weight_bytes(const pf_unet* u) { return u ? u->blob_floats * sizeof(float) : 0; }
int pf_unet_n_params(const pf_unet* u) { return u ? (int)u->params.size() : 0; }

int pf_unet_param_info(const pf_unet* u, int i, char* key_buf, size_t key_buf_len, int64_t shape[4], int* ndim) {
  PF_REQUIRE(u && i >= 0 && i < (int)u->params.size() && key_buf && shape && ndim, "pf_unet_param_info: bad arguments");
  const ParamSpec& ps = u->params[i];
  snprintf(key_buf, key_buf_len, "%s", ps.key.c_str());
  *ndim = (int)ps.shape.size();
  for (int d = 0; d < 4; ++d) shape[d] = d < *ndim ? ps.shape[d] : 1;
  return PF_OK;
}

int pf_unet_pack_param(pf_unet* u, const char* key, const float* src, const int64_t* shape, int ndim, void* host_blob) {
  PF_REQUIRE(u && key && src && shape && host_blob, "pf_unet_pack_param: null argument");
  auto it = u->index.find(key);
  if (it == u->index.end()) return set_error(PF_ENOTFOUND, "unexpected key '%s' (not a parameter of this UNet)", key);
  ParamSpec& ps = u->params[it->second];
  bool ok = ndim == (int)ps.shape.size();
  for (int d = 0; ok && d < ndim; ++d) ok = shape[d] == ps.shape[d];
  if (!ok) {
    std::string want, got;
    for (auto s : ps.shape) want += std::to_string(s) + ",";
    for (int d = 0; d < ndim; ++d) got += std::to_string(shape[d]) + ",";
    return set_error(PF_EINVAL, "size mismatch for '%s': expected [%s] got [%s]", key, want.c_str(), got.c_str());
  }
  if (pack_one(ps, src, (float*)host_blob) != PF_OK) return set_error(PF_EINVAL, X3_RANGE_MSG, key);
  ps.packed = true;
  return PF_OK;
}

int pf_unet_pack_missing(const pf_unet* u, char* buf, size_t buf_len) {
  if (!u) return set_error(PF_EINVAL, "null handle");
  int n = 0;
  for (const ParamSpec& ps : u->params)
    if (!ps.packed) {
      if (n == 0 && buf && buf_len) snprintf(buf, buf_len, "%s", ps.key.c_str());
      ++n;
    }
  return n;
}

int pf_unet_bind_weights(pf_unet* u, const void* dev_blob) {
  PF_REQUIRE(u && dev_blob, "pf_unet_bind_weights: null argument");
  PF_REQUIRE(((uintptr_t)dev_blob & 255) == 0, "pf_unet_bind_weights: blob must be 256-byte aligned");
  u->wdev = (const float*)dev_blob;
  return PF_OK;
}

size_t pf_unet_workspace_bytes(const pf_unet* u, int batch, int n_cond) {
  if (!u || batch <= 0 || n_cond <= 0) return 0;
  size_t p, t;
  plan_sizes(const_cast<pf_unet*>(u), batch, n_cond, &p, &t, nullptr);
  return p + t;
}

int pf_unet_n_launches(const pf_unet* u, int batch, int n_cond) {
  if (!u || batch <= 0 || n_cond <= 0) return 0;
  size_t p, t; int n = 0;
  plan_sizes(const_cast<pf_unet*>(u), batch, n_cond, &p, &t, &n);
  return n;
}

int pf_unet_forward(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch, int n_cond, float* eps,
                    void* workspace, size_t workspace_bytes, void* stream) {
  return pf_unet_forward_prepared(u, x, t, cond, batch, n_cond, nullptr, eps, workspace, workspace_bytes, stream);
}

static int forward_impl(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch, int n_cond, const pf_unet_prepared* prep,
                        float* eps, void* workspace, size_t workspace_bytes, void* stream, bool cfg_share) {
  PF_REQUIRE(u && x && t && cond && eps && workspace, "pf_unet_forward: null argument");
  PF_REQUIRE(!prep || !prep->time_table || prep->n_time_rows > 0, "pf_unet_forward: time table without rows");
  PF_REQUIRE(!prep || !prep->cross_bias || n_cond == 1, "pf_unet_forward: the collapsed cross-attention bias exists only for n_cond == 1");
  PF_REQUIRE(batch > 0 && n_cond > 0, "pf_unet_forward: batch and n_cond must be positive");
  PF_REQUIRE(!cfg_share || batch % 2 == 0, "pf_unet_forward_cfg: the batch is the two guidance halves (got %d)", batch);
  if (!u->wdev) return set_error(PF_ESTATE, "pf_unet_forward: weights not bound (call pf_unet_bind_weights)");
  PF_REQUIRE(n_cond == 1 || u->cfg.d_cond % 32 == 0, "pf_unet_forward: n_cond > 1 needs d_cond %% 32 == 0");
  PF_REQUIRE(((uintptr_t)workspace & 255) == 0, "pf_unet_forward: workspace must be 256-byte aligned");
  size_t p, tmp;
  plan_sizes(u, batch, n_cond, &p, &tmp, nullptr, false, false, cfg_share);
  if (workspace_bytes < p + tmp) return set_error(PF_EINVAL, "pf_unet_forward: workspace too small (%zu < %zu)", workspace_bytes, p + tmp);
  Ctx c{};
  c.u = u; c.s = (hipStream_t)stream; c.dry = false; c.base = (char*)workspace; c.temp_base = p;
  c.B = batch; c.n_cond = n_cond; c.W = u->wdev; c.rc = PF_OK; c.cfg_share = cfg_share;
  if (prep && prep->time_table) { c.has_time = true; c.prep_time = prep->time_table; c.prep_time_rows = prep->n_time_rows; }
  if (prep && prep->cross_bias && u->cross_total > 0) { c.has_cross = true; c.prep_cross = prep->cross_bias; }
  if (u->profiling) { u->n_prof = 0; u->pkind.clear(); u->pflops.clear(); }
  return run(u, c, x, t, cond, eps);
}

int pf_unet_forward_prepared(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch, int n_cond, const pf_unet_prepared* prep,
                             float* eps, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_impl(u, x, t, cond, batch, n_cond, prep, eps, workspace, workspace_bytes, stream, false);
}

int pf_unet_forward_cfg(pf_unet* u, const float* x, const int64_t* t, const float* cond, int batch2, int n_cond, const pf_unet_prepared* prep,
                        float* eps2, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_impl(u, x, t, cond, batch2, n_cond, prep, eps2, workspace, workspace_bytes, stream, true);
}

size_t pf_unet_workspace_bytes_cfg(const pf_unet* u, int batch2, int n_cond) {
  if (!u || batch2 <= 0 || batch2 % 2 || n_cond <= 0) return 0;
  size_t p, t;
  plan_sizes(const_cast<pf_unet*>(u), batch2, n_cond, &p, &t, nullptr, false, false, true);
  return p + t;
}

int pf_unet_n_launches_cfg(const pf_unet* u, int batch2, int n_cond, int has_time, int has_cross) {
  if (!u || batch2 <= 0 || batch2 % 2 || n_cond <= 0) return 0;
  size_t p, t; int n = 0;
  plan_sizes(const_cast<pf_unet*>(u), batch2, n_cond, &p, &t, &n, has_time != 0, has_cross != 0 && n_cond == 1, true);
  return n;
}

int pf_unet_time_bias_width(const pf_unet* u) { return u ? u->sum_emb : 0; }
int pf_unet_cross_bias_width(const pf_unet* u) { return u ? u->cross_total : 0; }

int pf_unet_prepare_time(pf_unet* u, int n_rows, float* table, void* scratch, size_t scratch_bytes, void* stream) {
  PF_REQUIRE(u && table && scratch && n_rows > 0, "pf_unet_prepare_time: bad arguments");
  if (!u->wdev) return set_error(PF_ESTATE, "pf_unet_prepare_time: weights not bound (call pf_unet_bind_weights)");
  PF_REQUIRE(scratch_bytes >= (size_t)n_rows * u->d_t * sizeof(float), "pf_unet_prepare_time: scratch too small (%zu < %zu)", scratch_bytes,
             (size_t)n_rows * u->d_t * sizeof(float));
  const float* W = u->wdev;
  float* tsilu = static_cast<float*>(scratch);
  // the same two launches forward issues per call (row r <- time-step value r): bit-identical to the unprepared path
  int rc = launch_time_embed(nullptr, W + u->te_w0, W + u->te_b0, W + u->te_w2, W + u->te_b2, tsilu, n_rows, u->cfg.channels, u->d_t, (hipStream_t)stream);
  if (rc != PF_OK) return rc;
  return launch_matvec(tsilu, u->d_t, W + u->emb_w, W + u->emb_b, table, u->sum_emb, n_rows, u->sum_emb, u->d_t, (hipStream_t)stream);
}

int pf_unet_prepare_cond(pf_unet* u, const float* cond, int batch, float* cross, void* scratch, size_t scratch_bytes, void* stream) {
  PF_REQUIRE(u && cond && cross && scratch && batch > 0, "pf_unet_prepare_cond: bad arguments");
  if (!u->wdev) return set_error(PF_ESTATE, "pf_unet_prepare_cond: weights not bound (call pf_unet_bind_weights)");
  PF_REQUIRE(u->cross_total > 0, "pf_unet_prepare_cond: this UNet has no transformer block");
  PF_REQUIRE(scratch_bytes >= (size_t)batch * u->cross_total * sizeof(float), "pf_unet_prepare_cond: scratch too small");
  Ctx c{};
  c.u = u; c.s = (hipStream_t)stream; c.dry = false; c.B = batch; c.n_cond = 1; c.W = u->wdev; c.rc = PF_OK;
  const bool prof = u->profiling;
  u->profiling = false;
  cross_bias_launches(u, c, cond, batch, cross, static_cast<float*>(scratch));
  u->profiling = prof;
  return c.rc;
}

int pf_unet_n_launches_prepared(const pf_unet* u, int batch, int n_cond, int has_time, int has_cross) {
  if (!u || batch <= 0 || n_cond <= 0) return 0;
  size_t p, t; int n = 0;
  plan_sizes(const_cast<pf_unet*>(u), batch, n_cond, &p, &t, &n, has_time != 0, has_cross != 0 && n_cond == 1);
  return n;
}

int pf_unet_set_option(pf_unet* u, int option, int value) {
  PF_REQUIRE(u && option >= 0 && option < PF_OPT_COUNT && value >= PF_OPT_AUTO && value <= PF_OPT_ON, "pf_unet_set_option: bad arguments");
  u->opt[option] = value;
  return PF_OK;
}
int pf_unet_track_absmax(pf_unet* u, void* device_word) {
  PF_REQUIRE(u, "null handle");
  u->amax_slot = device_word;
  return PF_OK;
}
int pf_unet_get_option(const pf_unet* u, int option) { return (u && option >= 0 && option < PF_OPT_COUNT) ? u->opt[option] : -2; }

int pf_unet_set_precision(pf_unet* u, int precision) {
  PF_REQUIRE(u && (precision == PF_PREC_F32 || precision == PF_PREC_BF16X3), "pf_unet_set_precision: bad arguments");
  u->precision = precision;
  return PF_OK;
}
int pf_unet_get_precision(const pf_unet* u) { return u ? u->precision : -1; }
int pf_x3_element(void) {
#ifdef PF_X3_F16
  return 1;
#else
  return 0;
#endif
}

int pf_unet_set_profiling(pf_unet* u, int enabled) {
  PF_REQUIRE(u, "null handle");
  u->profiling = enabled != 0;
  u->n_prof = 0; u->pkind.clear(); u->pflops.clear(); u->pdirect.clear();
  return PF_OK;
}

int pf_unet_profile_read(pf_unet* u, int* kind, float* ms, double* flops, int capacity) {
  PF_REQUIRE(u && kind && ms && flops, "pf_unet_profile_read: null argument");
  const int n = u->n_prof < capacity ? u->n_prof : capacity;
  for (int i = 0; i < n; ++i) {
    PF_CHECK_HIP(hipEventSynchronize(u->ev[(size_t)i * 2 + 1]));
    float v = 0.f;
    PF_CHECK_HIP(hipEventElapsedTime(&v, u->ev[(size_t)i * 2], u->ev[(size_t)i * 2 + 1]));
    kind[i] = u->pkind[i]; ms[i] = v; flops[i] = u->pflops[i];
  }
  return n;
}

int pf_unet_profile_read_direct(pf_unet* u, double* direct_flops, int capacity) {
  PF_REQUIRE(u && direct_flops, "pf_unet_profile_read_direct: null argument");
  const int n = u->n_prof < capacity ? u->n_prof : capacity;
  for (int i = 0; i < n; ++i) direct_flops[i] = u->pdirect[i];
  return n;
}

size_t pf_packed_gemm_weight_floats(int n, int k, int taps) { return pf_unet::gemm_floats(taps, k, n); }
int pf_pack_gemm_weight(const float* w, int n, int k, int taps, float* dst) {
  PF_REQUIRE(w && dst && n > 0 && k > 0 && k % 4 == 0 && (taps == 1 || taps == 9), "pf_pack_gemm_weight: bad arguments");
  memset(dst, 0, pf_unet::gemm_floats(taps, k, n) * sizeof(float));
  pack_gemm(dst, w, n, k, taps, (n + 63) / 64 * 64, 0);
  return PF_OK;
}

int pf_pack_gemm_weight_bf16x3(const float* w, int n, int k, int taps, void* dst) {
  PF_REQUIRE(w && dst && n > 0 && k > 0 && k % 8 == 0 && (taps == 1 || taps == 9), "pf_pack_gemm_weight_bf16x3: bad arguments");
  memset(dst, 0, pf_unet::gemm_floats(taps, k, n) * sizeof(float));
  PF_REQUIRE(pack_gemm_bf3(dst, w, n, k, taps, (n + 63) / 64 * 64, 0, nullptr), X3_RANGE_MSG, "pf_pack_gemm_weight_bf16x3");
  return PF_OK;
}

int pf_gn_scale_shift(const float* x0, int c0, const float* x1, int c1, int batch, int hw, int groups, float eps,
                      const float* gamma, const float* beta, float* scale, float* shift, void* scratch, size_t scratch_bytes,
                      void* stream) {
  return launch_gn_scale_shift(x0, c0, x1, c1, batch, hw, groups, eps, gamma, beta, scale, shift, scratch, scratch_bytes, (hipStream_t)stream);
}
int pf_pack_upfold_weight_bf16x3(const float* w, int n, int k, void* dst) {
  PF_REQUIRE(w && dst && n > 0 && k > 0 && k % 8 == 0, "pack_upfold: bad arguments");
  PF_REQUIRE(pack_upfold_bf3(dst, w, n, k, (n + 63) / 64 * 64), X3_RANGE_MSG, "pf_pack_upfold_weight_bf16x3");
  return PF_OK;
}
size_t pf_wino_weight_bytes(int n, int k) { return (n > 0 && k > 0) ? (size_t)16 * k * n * 4 : 0; }
int pf_pack_wino_weight_bf16x3(const float* w, int n, int k, void* dst) {
  PF_REQUIRE(w && dst && n > 0 && k > 0 && n % 64 == 0 && k % 16 == 0, "pack_wino: n must be a multiple of 64 and k of 16 (n=%d k=%d)", n, k);
  PF_REQUIRE(pack_wino_bf3(dst, w, n, k), X3_RANGE_MSG, "pf_pack_wino_weight_bf16x3");
  return PF_OK;
}
int pf_prmat2c_durations(const float* prmat2c, int n, int steps, int custom_round, int32_t* dur, void* stream) {
  return launch_prmat2c_durations(prmat2c, n, steps, custom_round, dur, (hipStream_t)stream);
}
int pf_mlp_geglu_fused(const float* x, int batch, int l, const float* ln_gamma, const float* ln_beta, float ln_eps,
                       const void* w1_bf16x3, const float* b1, const void* w2_bf16x3, const float* b2,
                       float* out, void* out_planes, void* stream) {
  return launch_mlp_fused(x, batch, l, ln_gamma, ln_beta, ln_eps, w1_bf16x3, b1, w2_bf16x3, b2, out, out_planes, (hipStream_t)stream);
}
int pf_mlp_geglu_proj_fused(const float* x, int batch, int l, const float* ln_gamma, const float* ln_beta, float ln_eps,
                            const void* w1_bf16x3, const float* b1, const void* w2_bf16x3, const float* b2,
                            const void* w3_bf16x3, const float* b3, const float* res3, float* out, float* stats3, void* stream) {
  if (!w3_bf16x3) return set_error(PF_EINVAL, "pf_mlp_geglu_proj_fused: null projection weight");
  return launch_mlp_fused(x, batch, l, ln_gamma, ln_beta, ln_eps, w1_bf16x3, b1, w2_bf16x3, b2, out, nullptr, (hipStream_t)stream, w3_bf16x3, b3,
                          res3, stats3);
}
int pf_ln_planes(const float* x, int rows, int c, float eps, const float* gamma, const float* beta, void* planes, void* stream) {
  return launch_ln_planes(x, rows, c, eps, gamma, beta, planes, (hipStream_t)stream);
}
int pf_ln_stats(const float* x, int rows, int c, float eps, float* mean, float* rstd, void* stream) {
  return launch_ln_stats(x, rows, c, eps, mean, rstd, (hipStream_t)stream);
}
int pf_conv_stats_tiles(const pf_conv_args* a) { return a ? conv_stats_tiles(*a) : 0; }
size_t pf_conv_splitk_ws_bytes(const pf_conv_args* a) { return a ? conv_splitk_ws_bytes(*a) : 0; }
int pf_gn_finalize_tiles(const float* stats0, int tiles0, int c0, const float* stats1, int tiles1, int c1, int batch, int hw,
                         int groups, float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  return launch_gn_finalize_tiles(stats0, tiles0, c0, stats1, tiles1, c1, batch, hw, groups, eps, gamma, beta, scale, shift,
                                  (hipStream_t)stream);
}
int pf_conv2d(const pf_conv_args* a, void* stream) {
  PF_REQUIRE(a, "pf_conv2d: null argument");
  return launch_conv(*a, (hipStream_t)stream);
}
size_t pf_attention_split_scratch_bytes(int batch, int n_heads, int l) {
  return (batch > 0 && n_heads > 0 && l > 0) ? attention_bf3_split_floats(batch, n_heads, l, nullptr) * sizeof(float) : 0;
}
int pf_attention_bf16x3_split(const void* qkv_planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, void* scratch, size_t scratch_bytes,
                              void* stream) {
  return launch_attention_bf3(qkv_planes, o, ldo, o_planes, batch, n_heads, l, 0, (hipStream_t)stream, static_cast<float*>(scratch), scratch_bytes / sizeof(float));
}
int pf_attention_bf16x3(const void* qkv_planes, float* o, int ldo, void* o_planes, int batch, int n_heads, int l, int form, void* stream) {
  return launch_attention_bf3(qkv_planes, o, ldo, o_planes, batch, n_heads, l, form, (hipStream_t)stream);
}
int pf_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, int batch,
                 int n_heads, int d_head, int lq, int lk, void* stream) {
  PF_REQUIRE(q && k && v && o, "pf_attention: null argument");
  return launch_attention(q, ldq, k, ldk, v, ldv, o, ldo, batch, n_heads, d_head, lq, lk, (hipStream_t)stream);
}

int pf_cfg_combine(const float* eps2, float scale, float* eps, size_t n, void* stream) { return launch_cfg_combine(eps2, scale, eps, n, (hipStream_t)stream); }
int pf_ddpm_step(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig, const float* mask,
                 const pf_ddpm_coef* c, float* x_out, size_t n, void* stream) {
  PF_REQUIRE(c, "pf_ddpm_step: null coefficients");
  return launch_ddpm_step(x, eps, noise_p, noise_q, orig, mask, *c, x_out, n, (hipStream_t)stream);
}
int pf_axpby(const float* x, const float* noise, float a, float b, float* out, size_t n, void* stream) { return launch_axpby(x, noise, a, b, out, n, (hipStream_t)stream); }
int pf_ddim_step(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise, const float* mask,
                 const pf_ddim_coef* c, float* x_out, size_t n, void* stream) {
  PF_REQUIRE(c, "pf_ddim_step: null coefficients");
  return launch_ddim_step(x, eps, noise, orig, orig_noise, mask, *c, x_out, n, (hipStream_t)stream);
}
int pf_randn(float* out, size_t n, uint64_t seed, uint64_t stream_id, uint64_t elem_offset, void* stream) {
  return launch_randn(out, n, seed, stream_id, elem_offset, (hipStream_t)stream);
}
int pf_ddpm_step_rng(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* c, uint64_t seed,
                     uint64_t draw_q, uint64_t draw_p, uint64_t elem_offset, float* x_out, size_t n, void* stream) {
  PF_REQUIRE(c, "pf_ddpm_step_rng: null coefficients");
  return launch_ddpm_step_rng(x, eps, orig, mask, c, nullptr, nullptr, seed, draw_q, draw_p, elem_offset, x_out, n, (hipStream_t)stream);
}
int pf_ddim_step_rng(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask, const pf_ddim_coef* c,
                     uint64_t seed, uint64_t draw, uint64_t elem_offset, float* x_out, size_t n, void* stream) {
  PF_REQUIRE(c, "pf_ddim_step_rng: null coefficients");
  return launch_ddim_step_rng(x, eps, orig, orig_noise, mask, c, nullptr, nullptr, seed, draw, elem_offset, x_out, n, (hipStream_t)stream);
}
int pf_ddpm_step_rng_dev(const float* x, const float* eps, const float* orig, const float* mask, const pf_ddpm_coef* table,
                         const pf_step_state* st, uint64_t seed, uint64_t elem_offset, float* x_out, size_t n, void* stream) {
  PF_REQUIRE(table && st, "pf_ddpm_step_rng_dev: null table / state");
  return launch_ddpm_step_rng(x, eps, orig, mask, nullptr, table, st, seed, 0, 0, elem_offset, x_out, n, (hipStream_t)stream);
}
int pf_ddim_step_rng_dev(const float* x, const float* eps, const float* orig, const float* orig_noise, const float* mask,
                         const pf_ddim_coef* table, const pf_step_state* st, uint64_t seed, uint64_t elem_offset, float* x_out, size_t n,
                         void* stream) {
  PF_REQUIRE(table && st, "pf_ddim_step_rng_dev: null table / state");
  return launch_ddim_step_rng(x, eps, orig, orig_noise, mask, nullptr, table, st, seed, 0, elem_offset, x_out, n, (hipStream_t)stream);
}
int pf_mfma_probe(float* sink, int iters, double* flops_out, void* stream) { return launch_mfma_probe(sink, iters, flops_out, (hipStream_t)stream); }
int pf_clock_probe(uint64_t* out2, void* stream) { return launch_clock_probe(reinterpret_cast<unsigned long long*>(out2), (hipStream_t)stream); }
int pf_step_state_set(pf_step_state* st, int64_t index, uint64_t draws, void* stream) { return launch_step_state_set(st, index, draws, (hipStream_t)stream); }
int pf_step_begin(const pf_step_state* st, const int32_t* time_steps, int64_t* t_out, int batch, void* stream) {
  return launch_step_begin(st, time_steps, t_out, batch, (hipStream_t)stream);
}
int pf_step_end(pf_step_state* st, int draws_used, void* stream) { return launch_step_end(st, draws_used, (hipStream_t)stream); }
int pf_randn_dev(float* out, size_t n, uint64_t seed, const pf_step_state* st, int slot, uint64_t elem_offset, void* stream) {
  return launch_randn_dev(out, n, seed, st, slot, elem_offset, (hipStream_t)stream);
}
int pf_ddpm_step_dev(const float* x, const float* eps, const float* noise_p, const float* noise_q, const float* orig, const float* mask,
                     const pf_ddpm_coef* table, const pf_step_state* st, float* x_out, size_t n, void* stream) {
  return launch_ddpm_step_dev(x, eps, noise_p, noise_q, orig, mask, table, st, x_out, n, (hipStream_t)stream);
}
int pf_ddim_step_dev(const float* x, const float* eps, const float* noise, const float* orig, const float* orig_noise, const float* mask,
                     const pf_ddim_coef* table, const pf_step_state* st, float* x_out, size_t n, void* stream) {
  return launch_ddim_step_dev(x, eps, noise, orig, orig_noise, mask, table, st, x_out, n, (hipStream_t)stream);
}

}  // extern "C"
