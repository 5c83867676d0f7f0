// encoders.hip - frozen condition encoders (run once per generation, before the step loop).
//   chord:   RnnEncoder(36 -> bi-GRU 512 -> mu 512)                (dl_modules/chord_enc.py:5-22)
//   texture: TextureEncoder(conv 4x12 -> ReLU -> pool -> fc1 -> fc2 -> bi-GRU 1024 -> mu 256)
//                                                                   (dl_modules/txt_enc.py:5-35)
// Only Normal(mu, .).mean is consumed downstream (models/model_sdf.py:99,157), so linear_var is
// accepted at pack time (checkpoint compatibility) and ignored.  Weights keep their torch
// row-major layout; every contraction is the batched mat-vec kernel (weights stream once per
// 8 samples), the recurrence is one mat-vec + one gate kernel per time step.
#include <string.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "pf_internal.h"

using namespace pf;

struct pf_encoder {
  int kind, input_dim, emb, hidden, z, nch;
  struct P { std::string key; std::vector<int64_t> shape; size_t off; bool packed; bool used; };
  std::vector<P> params;
  std::map<std::string, int> index;
  size_t blob_floats = 0;
  const float* wdev = nullptr;
  size_t add(const std::string& key, std::vector<int64_t> shape, bool used = true) {
    size_t n = 1; for (auto s : shape) n *= (size_t)s;
    size_t off = blob_floats;
    if (used) blob_floats += (n + 63) / 64 * 64;
    index[key] = (int)params.size();
    params.push_back(P{key, shape, off, false, used});
    return off;
  }
  size_t off(const std::string& key) const { return params[index.at(key)].off; }
};

extern "C" {

int pf_encoder_create(int kind, int input_dim, int emb_size, int hidden_dim, int z_dim, int num_channel, pf_encoder** out) {
  PF_REQUIRE(out && (kind == PF_ENC_CHORD || kind == PF_ENC_TEXTURE || kind == PF_ENC_PNOTREE), "pf_encoder_create: bad kind");
  PF_REQUIRE(hidden_dim > 0 && z_dim > 0, "pf_encoder_create: bad dims");
  std::unique_ptr<pf_encoder> e(new pf_encoder());
  e->kind = kind; e->input_dim = input_dim; e->emb = emb_size; e->hidden = hidden_dim; e->z = z_dim; e->nch = num_channel;
  int gru_in = input_dim;
  if (kind == PF_ENC_PNOTREE) {   // dl_modules/pianotree_enc.py:43-59
    PF_REQUIRE(input_dim > 5 && emb_size > 0 && num_channel > 0, "pf_encoder_create: pianotree encoder needs input_dim, emb_size and the note-GRU size");
    const int Hn = num_channel;
    e->add("note_embedding.weight", {emb_size, input_dim});
    e->add("note_embedding.bias", {emb_size});
    for (const char* sfx : {"", "_reverse"}) {
      e->add(std::string("enc_notes_gru.weight_ih_l0") + sfx, {3 * Hn, emb_size});
      e->add(std::string("enc_notes_gru.weight_hh_l0") + sfx, {3 * Hn, Hn});
      e->add(std::string("enc_notes_gru.bias_ih_l0") + sfx, {3 * Hn});
      e->add(std::string("enc_notes_gru.bias_hh_l0") + sfx, {3 * Hn});
    }
    for (const char* sfx : {"", "_reverse"}) {
      e->add(std::string("enc_time_gru.weight_ih_l0") + sfx, {3 * hidden_dim, 2 * Hn});
      e->add(std::string("enc_time_gru.weight_hh_l0") + sfx, {3 * hidden_dim, hidden_dim});
      e->add(std::string("enc_time_gru.bias_ih_l0") + sfx, {3 * hidden_dim});
      e->add(std::string("enc_time_gru.bias_hh_l0") + sfx, {3 * hidden_dim});
    }
    e->add("linear_mu.weight", {z_dim, 2 * hidden_dim});
    e->add("linear_mu.bias", {z_dim});
    e->add("linear_std.weight", {z_dim, 2 * hidden_dim}, false);
    e->add("linear_std.bias", {z_dim}, false);
    *out = e.release();
    return PF_OK;
  }
  if (kind == PF_ENC_TEXTURE) {
    PF_REQUIRE(num_channel > 0 && emb_size > 0, "pf_encoder_create: texture encoder needs num_channel and emb_size");
    e->add("cnn.0.weight", {num_channel, 1, 4, 12});
    e->add("cnn.0.bias", {num_channel});
    e->add("fc1.weight", {1000, num_channel * 29});
    e->add("fc1.bias", {1000});
    e->add("fc2.weight", {emb_size, 1000});
    e->add("fc2.bias", {emb_size});
    gru_in = emb_size;
  }
  for (const char* sfx : {"", "_reverse"}) {
    e->add(std::string("gru.weight_ih_l0") + sfx, {3 * hidden_dim, gru_in});
    e->add(std::string("gru.weight_hh_l0") + sfx, {3 * hidden_dim, hidden_dim});
    e->add(std::string("gru.bias_ih_l0") + sfx, {3 * hidden_dim});
    e->add(std::string("gru.bias_hh_l0") + sfx, {3 * hidden_dim});
  }
  e->add("linear_mu.weight", {z_dim, 2 * hidden_dim});
  e->add("linear_mu.bias", {z_dim});
  e->add("linear_var.weight", {z_dim, 2 * hidden_dim}, false);
  e->add("linear_var.bias", {z_dim}, false);
  *out = e.release();
  return PF_OK;
}

void pf_encoder_destroy(pf_encoder* e) { delete e; }
size_t pf_encoder_weight_bytes(const pf_encoder* e) { return e ? e->blob_floats * sizeof(float) : 0; }

int pf_encoder_pack_param(pf_encoder* e, const char* key, const float* src, const int64_t* shape, int ndim, void* host_blob) {
  PF_REQUIRE(e && key && src && shape && host_blob, "pf_encoder_pack_param: null argument");
  auto it = e->index.find(key);
  if (it == e->index.end()) return set_error(PF_ENOTFOUND, "unexpected key '%s' (not a parameter of this encoder)", key);
  pf_encoder::P& p = e->params[it->second];
  bool ok = ndim == (int)p.shape.size();
  size_t n = 1;
  for (int d = 0; ok && d < ndim; ++d) { ok = shape[d] == p.shape[d]; n *= (size_t)shape[d]; }
  if (!ok) return set_error(PF_EINVAL, "size mismatch for '%s'", key);
  if (p.used) memcpy((float*)host_blob + p.off, src, n * sizeof(float));
  p.packed = true;
  return PF_OK;
}

int pf_encoder_pack_missing(const pf_encoder* e, char* buf, size_t buf_len) {
  if (!e) return set_error(PF_EINVAL, "null handle");
  int n = 0;
  for (auto& p : e->params)
    if (!p.packed && p.used) {
      if (n == 0 && buf && buf_len) snprintf(buf, buf_len, "%s", p.key.c_str());
      ++n;
    }
  return n;
}

int pf_encoder_bind_weights(pf_encoder* e, const void* dev_blob) {
  PF_REQUIRE(e && dev_blob, "pf_encoder_bind_weights: null argument");
  e->wdev = (const float*)dev_blob;
  return PF_OK;
}

static size_t enc_ws_floats(const pf_encoder* e, int B, int T) {
  const size_t H = e->hidden;
  size_t f = 0;
  if (e->kind == PF_ENC_PNOTREE) {   // T = max_simu_note; B two-bar segments of 32 steps
    const size_t N = (size_t)B * 32, Hn = e->nch;
    f += N * T * e->emb + N + N * T * 3 * Hn + N * 3 * Hn + N * 2 * Hn;          // embedding, lengths, note-GRU gi / gh / [h_f | h_b]
    f += 2 * N * 3 * H + (size_t)B * 3 * H + (size_t)B * 2 * H;                  // time-GRU gi (both directions), gh, [h_f | h_b]
    return f + 16 * 64 + 1024;
  }
  if (e->kind == PF_ENC_TEXTURE) f += (size_t)B * e->nch * 8 * 29 + (size_t)B * 8 * 1000 + (size_t)B * 8 * e->emb;
  f += 2 * (size_t)B * T * 3 * H;  // gi forward / backward
  f += (size_t)B * 3 * H;          // gh
  f += (size_t)B * 2 * H;          // [h_f | h_b]
  return f + 1024;
}

size_t pf_encoder_workspace_bytes(const pf_encoder* e, int batch) {
  if (!e || batch <= 0) return 0;
  return enc_ws_floats(e, batch, e->kind == PF_ENC_TEXTURE ? 8 : (e->kind == PF_ENC_PNOTREE ? 32 : 64)) * sizeof(float);
}

int pf_encoder_forward(pf_encoder* e, const float* x, int batch, int n_step, float* mu, void* workspace, size_t workspace_bytes,
                       void* stream) {
  PF_REQUIRE(e && x && mu && workspace && batch > 0, "pf_encoder_forward: bad arguments");
  if (!e->wdev) return set_error(PF_ESTATE, "pf_encoder_forward: weights not bound");
  hipStream_t s = (hipStream_t)stream;
  const int B = batch, H = e->hidden;
  int T = n_step, gru_in = e->input_dim;
  PF_REQUIRE(e->kind == PF_ENC_TEXTURE || (T > 0 && T <= 64), "pf_encoder_forward: n_step must be in 1..64");
  if (e->kind == PF_ENC_PNOTREE) {
    // PianoTreeEncoder.forward (dl_modules/pianotree_enc.py:97-121): embed every grid row, run the bidirectional note GRU over the
    // (variable number of) notes of each of the B*32 time steps, then the bidirectional time GRU over the 32 steps, then linear_mu
    const int S = T, E = e->emb, Hn = e->nch, N = B * 32;
    PF_REQUIRE(S <= 32, "pf_encoder_forward: at most 32 simultaneous notes");
    PF_REQUIRE(workspace_bytes >= enc_ws_floats(e, B, S) * sizeof(float), "pf_encoder_forward: workspace too small");
    const float* W = e->wdev;
    float* ws = (float*)workspace;
    auto take = [&](size_t n) { float* p = ws; ws += (n + 63) / 64 * 64; return p; };
    float* emb = take((size_t)N * S * E);
    int* lens = reinterpret_cast<int*>(take(N));
    float* gin = take((size_t)N * S * 3 * Hn);
    float* ghn = take((size_t)N * 3 * Hn);
    float* hn = take((size_t)N * 2 * Hn);
    int rc = launch_pnotree_embed(x, W + e->off("note_embedding.weight"), W + e->off("note_embedding.bias"), emb, N * S, E, e->input_dim - 5, s);
    if (rc) return rc;
    rc = launch_pnotree_lengths(x, lens, N, S, e->input_dim - 5, s);
    if (rc) return rc;
    PF_CHECK_HIP(hipMemsetAsync(hn, 0, (size_t)N * 2 * Hn * sizeof(float), s));
    const char* sfx2[2] = {"", "_reverse"};
    for (int d = 0; d < 2; ++d) {
      const std::string sx = sfx2[d];
      rc = launch_matvec(emb, E, W + e->off("enc_notes_gru.weight_ih_l0" + sx), W + e->off("enc_notes_gru.bias_ih_l0" + sx), gin, 3 * Hn, N * S, 3 * Hn, E, s);
      if (rc) return rc;
      float* h = hn + d * Hn;
      for (int step = 0; step < S; ++step) {
        rc = launch_matvec(h, 2 * Hn, W + e->off("enc_notes_gru.weight_hh_l0" + sx), W + e->off("enc_notes_gru.bias_hh_l0" + sx), ghn, 3 * Hn, N, 3 * Hn, Hn, s);
        if (rc) return rc;
        rc = launch_gru_gates_masked(gin, ghn, h, 2 * Hn, N, Hn, S, lens, step, d, s);
        if (rc) return rc;
      }
    }
    float* git[2] = {take((size_t)N * 3 * H), nullptr};
    git[1] = take((size_t)N * 3 * H);
    float* ght = take((size_t)B * 3 * H);
    float* ht = take((size_t)B * 2 * H);
    PF_CHECK_HIP(hipMemsetAsync(ht, 0, (size_t)B * 2 * H * sizeof(float), s));
    for (int d = 0; d < 2; ++d) {
      const std::string sx = sfx2[d];
      rc = launch_matvec(hn, 2 * Hn, W + e->off("enc_time_gru.weight_ih_l0" + sx), W + e->off("enc_time_gru.bias_ih_l0" + sx), git[d], 3 * H, N, 3 * H, 2 * Hn, s);
      if (rc) return rc;
      float* h = ht + d * H;
      for (int step = 0; step < 32; ++step) {
        const int t = d == 0 ? step : 31 - step;
        rc = launch_matvec(h, 2 * H, W + e->off("enc_time_gru.weight_hh_l0" + sx), W + e->off("enc_time_gru.bias_hh_l0" + sx), ght, 3 * H, B, 3 * H, H, s);
        if (rc) return rc;
        rc = launch_gru_gates(git[d] + (size_t)t * 3 * H, 32 * 3 * H, ght, h, 2 * H, B, H, s);
        if (rc) return rc;
      }
    }
    return launch_matvec(ht, 2 * H, W + e->off("linear_mu.weight"), W + e->off("linear_mu.bias"), mu, e->z, B, e->z, 2 * H, s);
  }
  if (e->kind == PF_ENC_TEXTURE) { T = 8; gru_in = e->emb; }
  PF_REQUIRE(workspace_bytes >= enc_ws_floats(e, B, T) * sizeof(float), "pf_encoder_forward: workspace too small");
  const float* W = e->wdev;
  float* ws = (float*)workspace;
  auto take = [&](size_t n) { float* p = ws; ws += (n + 63) / 64 * 64; return p; };
  const float* seq = x;  // [B][T][gru_in]
  int rc = PF_OK;
  if (e->kind == PF_ENC_TEXTURE) {
    float* feat = take((size_t)B * e->nch * 8 * 29);
    float* a1 = take((size_t)B * 8 * 1000);
    float* em = take((size_t)B * 8 * e->emb);
    rc = launch_txt_frontend(x, W + e->off("cnn.0.weight"), W + e->off("cnn.0.bias"), feat, B, e->nch, s);
    if (rc) return rc;
    const int K1 = e->nch * 29;  // rows of the un-permuted [B,8,-1] view (txt_enc.py:27)
    rc = launch_matvec(feat, K1, W + e->off("fc1.weight"), W + e->off("fc1.bias"), a1, 1000, B * 8, 1000, K1, s);
    if (rc) return rc;
    rc = launch_matvec(a1, 1000, W + e->off("fc2.weight"), W + e->off("fc2.bias"), em, e->emb, B * 8, e->emb, 1000, s);
    if (rc) return rc;
    seq = em;
  }
  float* gi[2] = {take((size_t)B * T * 3 * H), nullptr};
  gi[1] = take((size_t)B * T * 3 * H);
  float* gh = take((size_t)B * 3 * H);
  float* hcat = take((size_t)B * 2 * H);
  PF_CHECK_HIP(hipMemsetAsync(hcat, 0, (size_t)B * 2 * H * sizeof(float), s));
  const char* sfx[2] = {"", "_reverse"};
  for (int d = 0; d < 2; ++d) {
    const std::string sx = sfx[d];
    rc = launch_matvec(seq, gru_in, W + e->off("gru.weight_ih_l0" + sx), W + e->off("gru.bias_ih_l0" + sx), gi[d], 3 * H, B * T, 3 * H, gru_in, s);
    if (rc) return rc;
    float* h = hcat + d * H;  // row stride 2H
    for (int step = 0; step < T; ++step) {
      const int t = d == 0 ? step : T - 1 - step;
      rc = launch_matvec(h, 2 * H, W + e->off("gru.weight_hh_l0" + sx), W + e->off("gru.bias_hh_l0" + sx), gh, 3 * H, B, 3 * H, H, s);
      if (rc) return rc;
      rc = launch_gru_gates(gi[d] + (size_t)t * 3 * H, T * 3 * H, gh, h, 2 * H, B, H, s);
      if (rc) return rc;
    }
  }
  return launch_matvec(hcat, 2 * H, W + e->off("linear_mu.weight"), W + e->off("linear_mu.bias"), mu, e->z, B, e->z, 2 * H, s);
}

}  // extern "C"
