// libmldhip: engine + C ABI (include/mldhip.h) for the MLD sampling hot path on MI355X (gfx950).
//
// Host-side structure (what the reference leaves to PyTorch/Lightning, rebuilt natively):
//   ParamTable   one HBM arena holding every weight the path reads, laid out per layer in execution
//                order with a uniform per-layer stride (lets one launch cover all 9 layers' tiny
//                cross-attention GEMMs via blockIdx.z);
//   Workspace    one HBM arena for activations, sized from (max_batch, max_frames) at create();
//   Schedule     DDIM tables (float32, as diffusers keeps them) + the time-MLP output for each of
//                the scheduler's timesteps, computed once at finalize (they depend on weights only);
//   sample()     ~2.2k kernel launches captured once per (B, Tmax, buffers) into a hipGraph and
//                replayed: the 50-step loop has no host work and no host<->device sync.
//
// Reference call stack being replaced: mld/models/modeltype/mld.py:216-265,290-360.
#include "../../include/mldhip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "kernels/attention.hpp"
#include "kernels/elementwise.hpp"
#include "kernels/gemm.hpp"
#include "kernels/novae.hpp"
#include "kernels/rt.hpp"
#include "kernels/tile32.hpp"
#include "kernels/fused_layer.hpp"

using namespace mld;

namespace {

std::string g_last_error;   // for failures before a handle exists

struct Param {
  std::string key;
  std::vector<int64_t> shape;
  size_t offset = 0;   // floats into the arena
  size_t numel = 0;
  bool loaded = false;
  int group = 0;       // 0 denoiser, 1 vae decoder, 2 dataset statistics
};

struct EncLayerP {   // TransformerEncoderLayer (cross_attention.py:236-272)
  const float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
};
struct DecLayerP {   // TransformerDecoderLayer (cross_attention.py:297-345)
  const float *in_w, *in_b, *out_w, *out_b;
  const float *cin_w, *cin_b, *cout_w, *cout_b;   // multihead_attn (only the V rows + out_proj are read)
  const float *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b;
};

// A captured sample() is independent of the caller's buffers: inputs are copied into engine-owned staging before
// the replay and outputs copied out after it (<= 17 MB of D2D copies, ~0.1 % of a batch), so a caller that allocates
// fresh output tensors on every call (as MLD.forward does) replays instead of re-capturing ~2 000 nodes.
struct GraphKey {
  int B, T;
  bool feats, joints;
  bool operator<(const GraphKey& o) const { return std::tie(B, T, feats, joints) < std::tie(o.B, o.T, o.feats, o.joints); }
};

}  // namespace

// One activation workspace.  cfg.max_in_flight of them share the weight arena: consecutive calls rotate through them,
// so calls issued on different streams overlap on the GPU (the reverse loop of one batch leaves most CUs idle most of
// the time: 1.6-1.8x throughput with 2-3 batches in flight, DESIGN.md §3 point 11).  A context is reused only after
// the stream of its new call has waited on the event recorded at the end of its previous call.
struct WsContext {
  float* ws = nullptr;
  int32_t *lens = nullptr, *lens2 = nullptr, *labels = nullptr;
  bool used = false;
#if !defined(MLDHIP_SIM)
  hipEvent_t done = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;
#endif
};

struct mldhip_engine {
  mldhip_config cfg;
  int device = 0;
  std::string err;
  bool finalized = false;
  bool group_ready[4] = {false, false, false, false};   // denoiser, vae decoder, mean/std, vae encoder

  // ---- parameters
  std::vector<Param> params;
  std::map<std::string, int> index;
  float* arena = nullptr;
  size_t arena_floats = 0;
  std::vector<EncLayerP> den;      // execution order
  std::vector<DecLayerP> dec;
  std::vector<EncLayerP> venc;     // VAE encoder layers (same layer type as the denoiser's)
  std::vector<DecLayerP> ndec;     // no-VAE variant: denoiser.decoder.layers.* (TransformerDecoder, cross_attention.py:195-233)
  size_t ndec_layer_stride = 0;
  float *TKV = nullptr, *XKV = nullptr, *TKV_one = nullptr;   // memory-token K|V per layer: time [L][n][2D], text [L][2*max_batch][2D]
  size_t dec_layer_stride = 0;     // floats between consecutive decoder layers' tensors

  // ---- schedule
  std::vector<int32_t> timesteps;
  std::vector<float> alphas_cumprod, betas;
  float final_alpha_cumprod = 1.f;

  // ---- workspace (the pointers below are those of the currently bound context)
  std::vector<WsContext> ctxs;
  std::vector<std::pair<float**, size_t>> carve;   // (member pointer, offset in floats) of every workspace buffer
  int cur_ctx = 0;
  unsigned next_ctx = 0;
  size_t ws_floats = 0;
  int32_t* lens_dev = nullptr;
  // denoiser
  float *X0, *Ha, *Hb, *H1, *S[8], *QKV, *AO, *FF, *lat, *T1, *temb0, *tmid, *text_bias, *t1_one, *temb0_one, *time_b2pe;
  // decode
  float *cv1, *cvec, *LNO, *feats_int, *joints_int, *zbuf;
  float *Po, *Pf, *Ps;   // denoiser split-K slabs: out-proj [1], FFN2 [4], skip-linear [2], each [6*max_batch][256]
  unsigned long long* trace_buf = nullptr;   // measurement only (mldhip_profile_trace)
  unsigned long long* trace_on = nullptr;    // non-null while a traced launch is being built
  float* WskelP = nullptr;   // skel_embedding.weight padded to [D][KP]
  int32_t* labels_dev = nullptr; // action labels of the CFG batch [2*max_batch] (uncond half first, ignored there)
  int32_t* lens2_dev = nullptr;  // lengths + 2 (encoder key-padding mask incl. the two distribution tokens)
  std::vector<int32_t> lens2_host;
  float *text_in = nullptr, *lat_in = nullptr;   // graph staging of the caller's inputs
  float* TP;             // text projection rows [2*max_batch][256] (+pe[2]), gathered per chain
  bool fused_ffn = false; // MLDHIP_FUSED_FFN=1: linear1+GELU+linear2 in one launch (measured slower: DESIGN.md §3 point 9)
  int t32_kh = 1;        // MLDHIP_T32_KH=2: tile32 kernels pass K through LDS in two pieces (two workgroups per CU)
  bool tile16 = true;    // MLDHIP_TILE16=0 disables the 16-row K-split tiles (A/B runs)
  int nchains = 1;       // independent sub-batch chains of the reverse loop (parallel graph branches)

  int launches[3] = {0, 0, 0};
  int phase = 0;

#if !defined(MLDHIP_SIM)
  hipStream_t cap_stream = nullptr;
  hipStream_t side[7] = {};
  hipEvent_t ev_fork = nullptr, ev_join[7] = {};
#endif

  int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

namespace {

using E = mldhip_engine;

#define HIP_TRY(e, call)                                                                       \
  do {                                                                                         \
    hipError_t _s = (call);                                                                    \
    if (_s != hipSuccess) return (e)->fail(MLDHIP_EHIP, "%s failed: %s", #call, hipGetErrorString(_s)); \
  } while (0)

void bind_context(E* e, int k) {
  WsContext& x = e->ctxs[k];
  for (auto& cv : e->carve) *cv.first = x.ws + cv.second;
  e->lens_dev = x.lens; e->lens2_dev = x.lens2; e->labels_dev = x.labels;
  e->cur_ctx = k;
}

// Scope of one workspace-using call on `stream`: picks the next context round-robin, orders the stream behind the
// context's previous user, binds its buffers; on exit records the context's "done" event on the stream.
struct CtxUse {
  E* e;
  hipStream_t stream;
  int rc = 0;
  CtxUse(E* e_, hipStream_t s) : e(e_), stream(s) {
    const int k = int(e->next_ctx++ % e->ctxs.size());
#if !defined(MLDHIP_SIM)
    WsContext& x = e->ctxs[k];
    if (x.used && e->ctxs.size() > 1) {
      hipError_t st = hipStreamWaitEvent(stream, x.done, 0);
      if (st != hipSuccess) rc = e->fail(MLDHIP_EHIP, "hipStreamWaitEvent(context): %s", hipGetErrorString(st));
    }
#endif
    bind_context(e, k);
  }
  ~CtxUse() {
#if !defined(MLDHIP_SIM)
    WsContext& x = e->ctxs[e->cur_ctx];
    if (e->ctxs.size() > 1) (void)hipEventRecord(x.done, stream);
    x.used = true;
#endif
  }
};

bool is_action(const E* e) { return e->cfg.condition == MLDHIP_COND_ACTION; }
bool is_actor(const E* e) { return e->cfg.vae_arch == MLDHIP_VAE_ACTOR; }
int time_width(const E* e) { return is_action(e) ? e->cfg.latent_dim : e->cfg.text_dim; }   // mld_denoiser.py:57-77
bool is_novae(const E* e) { return e->cfg.vae_arch == MLDHIP_VAE_NONE; }
bool is_ddpm(const E* e) { return e->cfg.scheduler_type == MLDHIP_SCHED_DDPM; }
int novae_kp(const E* e) { return (e->cfg.nfeats + 127) / 128 * 128; }   // feature width padded to 4 K chunks (263 -> 384)
int vae_layers(const E* e) { return is_actor(e) ? (e->cfg.vae_num_layers > 0 ? e->cfg.vae_num_layers : e->cfg.num_layers) : e->cfg.num_layers; }
std::string actor_layer(int i) { return "vae.decoder.seqTransDecoder.layers." + std::to_string(i); }

constexpr size_t kAlign = 64;   // floats
size_t align_up(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

std::vector<std::string> block_names(int num_block) {
  std::vector<std::string> v;
  if (num_block < 0) return v;
  for (int i = 0; i < num_block; ++i) v.push_back("input_blocks." + std::to_string(i));
  v.push_back("middle_block");
  for (int i = 0; i < num_block; ++i) v.push_back("output_blocks." + std::to_string(i));
  return v;
}

size_t add_param(E* e, const std::string& key, std::vector<int64_t> shape) {
  Param p;
  p.key = key;
  p.shape = shape;
  p.numel = 1;
  for (auto s : shape) p.numel *= size_t(s);
  p.offset = e->arena_floats;
  const bool enc = key.rfind("vae.encoder.", 0) == 0 || key.rfind("vae.skel_embedding.", 0) == 0 ||
                   key.rfind("vae.global_motion_token", 0) == 0 || key.rfind("vae.query_pos_encoder.", 0) == 0;
  p.group = key.rfind("denoiser.", 0) == 0 ? 0 : enc ? 3 : key.rfind("vae.", 0) == 0 ? 1 : 2;
  e->arena_floats += align_up(p.numel);
  e->index[key] = int(e->params.size());
  e->params.push_back(p);
  return p.offset;
}

// Declares every tensor the sampling path reads, in execution order.
void declare_params(E* e) {
  const auto& c = e->cfg;
  const int64_t D = c.latent_dim, F = c.ff_size, TD = c.text_dim, NF = c.nfeats;
  const int nb = (c.num_layers - 1) / 2;
  auto mha = [&](const std::string& p) {
    add_param(e, p + ".in_proj_weight", {3 * D, D});
    add_param(e, p + ".in_proj_bias", {3 * D});
    add_param(e, p + ".out_proj.weight", {D, D});
    add_param(e, p + ".out_proj.bias", {D});
  };
  auto lin = [&](const std::string& p, int64_t o, int64_t i) {
    add_param(e, p + ".weight", {o, i});
    add_param(e, p + ".bias", {o});
  };
  auto norm = [&](const std::string& p) {
    add_param(e, p + ".weight", {D});
    add_param(e, p + ".bias", {D});
  };
  if (is_novae(e)) {
    // diffusion-only denoiser (mld_denoiser.py:50-53,57-68,88-91,120-133): no VAE tensors at all
    lin("denoiser.pose_embd", D, NF);
    lin("denoiser.pose_proj", NF, D);
    lin("denoiser.time_embedding.linear_1", D, TD);
    lin("denoiser.time_embedding.linear_2", D, D);
    lin("denoiser.emb_proj.1", D, TD);
    add_param(e, "denoiser.query_pos.pe", {500, 1, D});
    add_param(e, "denoiser.mem_pos.pe", {500, 1, D});
    size_t first = 0, second = 0;
    for (int i = 0; i < c.num_layers; ++i) {
      std::string p = "denoiser.decoder.layers." + std::to_string(i);
      size_t start = e->arena_floats;
      mha(p + ".self_attn");
      mha(p + ".multihead_attn");
      lin(p + ".linear1", F, D);
      lin(p + ".linear2", D, F);
      norm(p + ".norm1");
      norm(p + ".norm2");
      norm(p + ".norm3");
      if (i == 0) first = start;
      if (i == 1) second = start;
    }
    e->ndec_layer_stride = second - first;
    norm("denoiser.decoder.norm");
    add_param(e, "mean", {NF});
    add_param(e, "std", {NF});
    return;
  }
  // denoiser (mld_denoiser.py:40-133)
  lin("denoiser.time_embedding.linear_1", D, time_width(e));
  lin("denoiser.time_embedding.linear_2", D, D);
  if (is_action(e)) add_param(e, "denoiser.emb_proj.action_embedding", {(int64_t)c.nclasses, D});   // EmbedAction
  else lin("denoiser.emb_proj.1", D, TD);
  add_param(e, "denoiser.query_pos.pe", {500, 1, D});
  for (auto& b : block_names(nb)) {
    std::string p = "denoiser.encoder." + b;
    mha(p + ".self_attn");
    lin(p + ".linear1", F, D);
    lin(p + ".linear2", D, F);
    norm(p + ".norm1");
    norm(p + ".norm2");
  }
  for (int i = 0; i < nb; ++i) lin("denoiser.encoder.linear_blocks." + std::to_string(i), D, 2 * D);
  norm("denoiser.encoder.norm");
  size_t first = 0, second = 0;
  int li = 0;
  if (is_actor(e)) {
    // ActorVae decoder (actor_vae.py:176-207): sinusoidal PE buffer, stock decoder layers, final_layer.  Its
    // encoder (training / reconstruction only) is not on the sampling path: vae.encoder.* keys are ignored.
    add_param(e, "vae.decoder.sequence_pos_encoding.pe", {5000, 1, D});
    for (int i = 0; i < vae_layers(e); ++i) {
      std::string p = actor_layer(i);
      size_t start = e->arena_floats;
      mha(p + ".self_attn");
      mha(p + ".multihead_attn");
      lin(p + ".linear1", F, D);
      lin(p + ".linear2", D, F);
      norm(p + ".norm1");
      norm(p + ".norm2");
      norm(p + ".norm3");
      if (i == 0) first = start;
      if (i == 1) second = start;
    }
    e->dec_layer_stride = second - first;
    lin("vae.decoder.final_layer", NF, D);
    // ActorVae encoder (actor_vae.py:84-175) -- optional weight group, like MldVae's: [mu_token | logvar_token] are
    // declared back to back so that together they form the [2][D] token block the token-assembly kernel expects
    add_param(e, "vae.encoder.mu_token", {D});
    add_param(e, "vae.encoder.logvar_token", {D});
    add_param(e, "vae.encoder.sequence_pos_encoding.pe", {5000, 1, D});
    lin("vae.encoder.skel_embedding", D, NF);
    for (int i = 0; i < vae_layers(e); ++i) {
      std::string p = "vae.encoder.seqTransEncoder.layers." + std::to_string(i);
      mha(p + ".self_attn");
      lin(p + ".linear1", F, D);
      lin(p + ".linear2", D, F);
      norm(p + ".norm1");
      norm(p + ".norm2");
    }
    add_param(e, "mean", {NF});
    add_param(e, "std", {NF});
    return;
  }
  // VAE decoder (mld_vae.py:85-112)
  add_param(e, "vae.query_pos_decoder.pe", {500, 1, D});
  for (auto& b : block_names(nb)) {
    std::string p = "vae.decoder." + b;
    size_t start = e->arena_floats;
    mha(p + ".self_attn");
    mha(p + ".multihead_attn");
    lin(p + ".linear1", F, D);
    lin(p + ".linear2", D, F);
    norm(p + ".norm1");
    norm(p + ".norm2");
    norm(p + ".norm3");
    if (li == 0) first = start;
    if (li == 1) second = start;
    ++li;
  }
  e->dec_layer_stride = second - first;
  for (int i = 0; i < nb; ++i) lin("vae.decoder.linear_blocks." + std::to_string(i), D, 2 * D);
  norm("vae.decoder.norm");
  lin("vae.final_layer", NF, D);
  // VAE encoder (mld_vae.py:75-83,108-111) -- scope row 8f.1; an optional weight group
  add_param(e, "vae.global_motion_token", {2 * (int64_t)c.latent_size, D});
  add_param(e, "vae.query_pos_encoder.pe", {500, 1, D});
  lin("vae.skel_embedding", D, NF);
  for (auto& b : block_names(nb)) {
    std::string p = "vae.encoder." + b;
    mha(p + ".self_attn");
    lin(p + ".linear1", F, D);
    lin(p + ".linear2", D, F);
    norm(p + ".norm1");
    norm(p + ".norm2");
  }
  for (int i = 0; i < nb; ++i) lin("vae.encoder.linear_blocks." + std::to_string(i), D, 2 * D);
  norm("vae.encoder.norm");
  add_param(e, "mean", {NF});
  add_param(e, "std", {NF});
}

const float* P(E* e, const std::string& key) { return e->arena + e->params[e->index.at(key)].offset; }

DecLayerP bind_dec_layer(E* e, const std::string& p) {
  DecLayerP L;
  L.in_w = P(e, p + ".self_attn.in_proj_weight"); L.in_b = P(e, p + ".self_attn.in_proj_bias");
  L.out_w = P(e, p + ".self_attn.out_proj.weight"); L.out_b = P(e, p + ".self_attn.out_proj.bias");
  L.cin_w = P(e, p + ".multihead_attn.in_proj_weight"); L.cin_b = P(e, p + ".multihead_attn.in_proj_bias");
  L.cout_w = P(e, p + ".multihead_attn.out_proj.weight"); L.cout_b = P(e, p + ".multihead_attn.out_proj.bias");
  L.l1_w = P(e, p + ".linear1.weight"); L.l1_b = P(e, p + ".linear1.bias");
  L.l2_w = P(e, p + ".linear2.weight"); L.l2_b = P(e, p + ".linear2.bias");
  L.n1_w = P(e, p + ".norm1.weight"); L.n1_b = P(e, p + ".norm1.bias");
  L.n2_w = P(e, p + ".norm2.weight"); L.n2_b = P(e, p + ".norm2.bias");
  L.n3_w = P(e, p + ".norm3.weight"); L.n3_b = P(e, p + ".norm3.bias");
  return L;
}

void bind_layers(E* e) {
  const int nb = (e->cfg.num_layers - 1) / 2;
  e->den.clear();
  e->dec.clear();
  e->ndec.clear();
  if (is_novae(e)) {
    for (int i = 0; i < e->cfg.num_layers; ++i) e->ndec.push_back(bind_dec_layer(e, "denoiser.decoder.layers." + std::to_string(i)));
    return;
  }
  for (auto& b : block_names(nb)) {
    std::string p = "denoiser.encoder." + b;
    EncLayerP L;
    L.in_w = P(e, p + ".self_attn.in_proj_weight"); L.in_b = P(e, p + ".self_attn.in_proj_bias");
    L.out_w = P(e, p + ".self_attn.out_proj.weight"); L.out_b = P(e, p + ".self_attn.out_proj.bias");
    L.l1_w = P(e, p + ".linear1.weight"); L.l1_b = P(e, p + ".linear1.bias");
    L.l2_w = P(e, p + ".linear2.weight"); L.l2_b = P(e, p + ".linear2.bias");
    L.n1_w = P(e, p + ".norm1.weight"); L.n1_b = P(e, p + ".norm1.bias");
    L.n2_w = P(e, p + ".norm2.weight"); L.n2_b = P(e, p + ".norm2.bias");
    e->den.push_back(L);
  }
  e->venc.clear();
  std::vector<std::string> venc_names;
  if (is_actor(e)) for (int i = 0; i < vae_layers(e); ++i) venc_names.push_back("vae.encoder.seqTransEncoder.layers." + std::to_string(i));
  else for (auto& b : block_names(nb)) venc_names.push_back("vae.encoder." + b);
  for (auto& p : venc_names) {
    EncLayerP L;
    L.in_w = P(e, p + ".self_attn.in_proj_weight"); L.in_b = P(e, p + ".self_attn.in_proj_bias");
    L.out_w = P(e, p + ".self_attn.out_proj.weight"); L.out_b = P(e, p + ".self_attn.out_proj.bias");
    L.l1_w = P(e, p + ".linear1.weight"); L.l1_b = P(e, p + ".linear1.bias");
    L.l2_w = P(e, p + ".linear2.weight"); L.l2_b = P(e, p + ".linear2.bias");
    L.n1_w = P(e, p + ".norm1.weight"); L.n1_b = P(e, p + ".norm1.bias");
    L.n2_w = P(e, p + ".norm2.weight"); L.n2_b = P(e, p + ".norm2.bias");
    e->venc.push_back(L);
  }
  std::vector<std::string> dec_names;
  if (is_actor(e)) for (int i = 0; i < vae_layers(e); ++i) dec_names.push_back(actor_layer(i));
  else for (auto& b : block_names(nb)) dec_names.push_back("vae.decoder." + b);
  for (auto& p : dec_names) {
    DecLayerP L;
    L.in_w = P(e, p + ".self_attn.in_proj_weight"); L.in_b = P(e, p + ".self_attn.in_proj_bias");
    L.out_w = P(e, p + ".self_attn.out_proj.weight"); L.out_b = P(e, p + ".self_attn.out_proj.bias");
    L.cin_w = P(e, p + ".multihead_attn.in_proj_weight"); L.cin_b = P(e, p + ".multihead_attn.in_proj_bias");
    L.cout_w = P(e, p + ".multihead_attn.out_proj.weight"); L.cout_b = P(e, p + ".multihead_attn.out_proj.bias");
    L.l1_w = P(e, p + ".linear1.weight"); L.l1_b = P(e, p + ".linear1.bias");
    L.l2_w = P(e, p + ".linear2.weight"); L.l2_b = P(e, p + ".linear2.bias");
    L.n1_w = P(e, p + ".norm1.weight"); L.n1_b = P(e, p + ".norm1.bias");
    L.n2_w = P(e, p + ".norm2.weight"); L.n2_b = P(e, p + ".norm2.bias");
    L.n3_w = P(e, p + ".norm3.weight"); L.n3_b = P(e, p + ".norm3.bias");
    e->dec.push_back(L);
  }
}

// DDIM tables, float32 throughout like diffusers (third-party, parity unpinned).
void build_schedule(E* e) {
  const auto& c = e->cfg;
  const int N = c.num_train_timesteps;
  const float start = sqrtf(c.beta_start), stop = sqrtf(c.beta_end);
  const float step = (stop - start) / float(N - 1);
  e->alphas_cumprod.resize(N);
  e->betas.resize(N);
  float prod = 1.f;
  for (int i = 0; i < N; ++i) {
    float y = (i == N - 1) ? stop : float(i) * step + start;
    float beta = y * y;
    e->betas[i] = beta;
    prod = prod * (1.0f - beta);
    e->alphas_cumprod[i] = prod;
  }
  e->final_alpha_cumprod = c.set_alpha_to_one ? 1.0f : e->alphas_cumprod[0];
  const int n = c.num_inference_steps, ratio = N / n;
  e->timesteps.resize(n);
  // DDIM: steps_offset shifts the grid (scheduler.yaml:14); DDPM.set_timesteps has no offset
  for (int i = 0; i < n; ++i) e->timesteps[i] = (n - 1 - i) * ratio + (is_ddpm(e) ? 0 : c.steps_offset);
}

// DDPM ancestral-step coefficients, variance_type fixed_small (third party, parity unpinned; float32 like diffusers).
DdpmCoef ddpm_coef(const E* e, int t) {
  const int prev = t - e->cfg.num_train_timesteps / e->cfg.num_inference_steps;
  const float ab_t = e->alphas_cumprod[t], ab_p = prev >= 0 ? e->alphas_cumprod[prev] : 1.0f;
  const bool unit = e->cfg.num_train_timesteps == e->cfg.num_inference_steps;   // ratio 1: table values (see oracle DDPMSchedule)
  const float a_t = unit ? 1.0f - e->betas[t] : ab_t / ab_p, b_t = unit ? e->betas[t] : 1.0f - a_t;
  const float bp_t = 1.0f - ab_t, bp_p = 1.0f - ab_p;
  DdpmCoef k;
  k.sqrt_ab = sqrtf(ab_t);
  k.sqrt_1mab = sqrtf(bp_t);
  k.c_x0 = sqrtf(ab_p) * b_t / bp_t;
  k.c_x = sqrtf(a_t) * bp_p / bp_t;
  k.sigma = t > 0 ? sqrtf(fmaxf(bp_p / bp_t * b_t, 1e-20f)) : 0.0f;
  return k;
}

DdimCoef ddim_coef(const E* e, int t) {
  const auto& c = e->cfg;
  const int prev = t - c.num_train_timesteps / c.num_inference_steps;
  const float at = e->alphas_cumprod[t];
  const float ap = prev >= 0 ? e->alphas_cumprod[prev] : e->final_alpha_cumprod;
  return DdimCoef{sqrtf(at), sqrtf(1.0f - at), sqrtf(ap), sqrtf(1.0f - ap)};
}

// get_timestep_embedding(flip_sin_to_cos=True, freq_shift=0) (embeddings.py:245-285) for one t.
void timestep_sincos(float t, int dim, float* out) {
  const int half = dim / 2;
  const float neg_log = float(-std::log(10000.0));
  for (int i = 0; i < half; ++i) {
    float expo = neg_log * float(i);
    expo = expo / float(half);
    const float ang = t * expf(expo);
    out[i] = cosf(ang);
    out[half + i] = sinf(ang);
  }
}

// ------------------------------------------------------------------------------------ launches

struct Ctx {
  E* e;
  hipStream_t stream;
  int rc = 0;
};

void count(Ctx& c) { c.e->launches[c.e->phase]++; }

int check_launch(Ctx& c, const char* what) {
#if !defined(MLDHIP_SIM)
  hipError_t s = hipGetLastError();
  if (s != hipSuccess && c.rc == 0) c.rc = c.e->fail(MLDHIP_EHIP, "launch %s: %s", what, hipGetErrorString(s));
#endif
  (void)what;
  return c.rc;
}

// Tile configurations.  "small" targets the latency-bound denoiser (M = 6B rows): one 16x16 tile per
// wave so a GEMM spreads over as many SIMDs as possible; "large" targets the MFMA-bound decoder.
int g_small_m = 256;         // MLDHIP_SMALL_M: row count up to which the 16x64 one-tile-per-wave shape is used (tiny one-off GEMMs)
bool g_gemm8 = true;         // MLDHIP_GEMM8=0: the 4-wave variants of the staged fp32 GEMM tiles (A/B runs)
bool g_staged_gemm = true;   // MLDHIP_GEMM=direct selects the first-version register-direct main loop (A/B runs)

// staged (LDS, prefetch ring) launch of one tile shape; K / 32 is a template parameter
template <int WM, int WN, int MREP, int NREP, bool LN, int PREC>
void launch_staged(Ctx& c, const GemmArgs& a, dim3 grid) {
  const int kcs = (a.K1 + a.K2) / 32;
  constexpr int lds = gemm_lds_bytes<WM, WN, MREP, NREP>();
  if (c.e->trace_on) {   // measurement build of the same kernel (K = 256 shapes only)
    GemmArgs t = a;
    t.trace = c.e->trace_on;
    if (kcs == 8) { MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 8, true>), grid, dim3(WM * WN * 64), lds, c.stream, t); }
    else if (kcs == 32) { MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 32, true>), grid, dim3(WM * WN * 64), lds, c.stream, t); }
    return;
  }
  switch (kcs) {
    case 8: MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 8>), grid, dim3(WM * WN * 64), lds, c.stream, a); break;
    case 12:   // K = 384: the 263-wide motion features padded to the chunk pipeline (pose_embd of the no-VAE denoiser)
      if constexpr (!LN && PREC == 0) { MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 12>), grid, dim3(WM * WN * 64), lds, c.stream, a); }
      else c.rc = c.e->fail(MLDHIP_EINVAL, "staged GEMM: K=384 is built for the plain fp32 tile only");
      break;
    case 16: MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 16>), grid, dim3(WM * WN * 64), lds, c.stream, a); break;
    case 32: MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 32>), grid, dim3(WM * WN * 64), lds, c.stream, a); break;
    default: c.rc = c.e->fail(MLDHIP_EINVAL, "staged GEMM: K=%d not in {256,384,512,1024}", a.K1 + a.K2);
  }
}
void gemm(Ctx& c, const GemmArgs& a, int nz = 1) {
  const int K = a.K1 + a.K2;
  const bool small = a.M <= g_small_m || (K != 256 && K != 384 && K != 512 && K != 1024);
  const bool x3 = c.e->cfg.precision >= MLDHIP_PREC_BF16X3_DECODE && c.e->phase == 1 && K != 384;   // decoder GEMMs only
  if (small) {
    dim3 grid((a.M + 15) / 16, (a.N + 63) / 64, nz);
    MLD_LAUNCH((gemm_kernel<1, 4, 1, 1, false>), grid, dim3(256), 0, c.stream, a);
  } else if (!g_staged_gemm) {
    dim3 grid((a.M + 63) / 64, (a.N + 127) / 128, nz);
    MLD_LAUNCH((gemm_kernel<2, 2, 2, 4, false>), grid, dim3(256), 0, c.stream, a);
  } else {
    dim3 grid((a.M + 63) / 64, (a.N + 127) / 128, nz);
    if (x3 && g_gemm8) launch_staged<2, 4, 2, 2, false, 1>(c, a, grid);
    else if (x3) launch_staged<2, 2, 2, 4, false, 1>(c, a, grid);
    else if (g_gemm8) launch_staged<2, 4, 2, 2, false, 0>(c, a, grid);   // same 64x128 tile on 8 waves (2 per SIMD)
    else launch_staged<2, 2, 2, 4, false, 0>(c, a, grid);
  }
  count(c);
  check_launch(c, "gemm");
}

void gemm_ln(Ctx& c, const GemmArgs& a) {   // N == 256; full rows per workgroup (32 x 256 tile)
  dim3 grid((a.M + 31) / 32, 1, 1);
  const bool x3 = c.e->cfg.precision >= MLDHIP_PREC_BF16X3_DECODE && c.e->phase == 1;
  if (!g_staged_gemm) {
    MLD_LAUNCH((gemm_kernel<1, 4, 2, 4, true>), grid, dim3(256), 0, c.stream, a);
  } else if (x3 && g_gemm8) {
    launch_staged<2, 4, 2, 4, true, 1>(c, a, dim3((a.M + 63) / 64, 1, 1));
  } else if (x3) {
    launch_staged<1, 4, 2, 4, true, 1>(c, a, grid);
  } else if (g_gemm8) {
    launch_staged<2, 4, 2, 4, true, 0>(c, a, dim3((a.M + 63) / 64, 1, 1));   // 64 x 256 tile on 8 waves
  } else {
    launch_staged<1, 4, 2, 4, true, 0>(c, a, grid);
  }
  count(c);
  check_launch(c, "gemm_ln");
}

GemmArgs lin_args(const float* A, int lda, int K, const float* W, const float* b, float* Y, int ldy, int M, int N) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.K1 = K; g.W = W; g.ldw = K; g.bias = b; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N;
  return g;
}

// ---- denoiser layer pipeline on the tile32 kernels (4 launches per encoder layer) ----------------
constexpr int t32_lds_bytes(int mt, int kh = 1) { return (mt + 64) * (256 / kh + 4) * 4; }

void tile32(Ctx& c, const Tile32Args& a_, int nz) {
  Tile32Args a = a_;
  a.trace = c.e->trace_on;
  // 16-row K-split tiles for the narrow (N = 256) GEMMs: more workgroups, fewer bytes and MFMAs per CU
  const bool mt16 = a.N <= 256 && ((a.M + 15) / 16) * ((a.N + 63) / 64) * nz <= 256 && c.e->tile16;
  const int mt = mt16 ? 16 : 32;
  dim3 grid((a.M + mt - 1) / mt, (a.N + 63) / 64, nz);
  const int ns = a.src[0].attn_R > 0 ? 0 : a.src[0].nsplit;
#define MLD_T32(MT, NS)                                                                                          \
  do {                                                                                                           \
    if (a.trace) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, true>), grid, dim3(512), t32_lds_bytes(MT), c.stream, a); }  \
    else if (c.e->t32_kh == 2) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, false, 2>), grid, dim3(512), t32_lds_bytes(MT, 2), c.stream, a); } \
    else { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, false>), grid, dim3(512), t32_lds_bytes(MT), c.stream, a); }         \
  } while (0)
#define MLD_T32_NS(MT)                                                                                           \
  switch (ns) {                                                                                                  \
    case 0: MLD_T32(MT, 0); break;                                                                               \
    case 1: MLD_T32(MT, 1); break;                                                                               \
    case 2: MLD_T32(MT, 2); break;                                                                               \
    case 4: MLD_T32(MT, 4); break;                                                                               \
    case 8: MLD_T32(MT, 8); break;                                                                               \
    default: c.rc = c.e->fail(MLDHIP_EINVAL, "tile32: unsupported slab count %d", ns); return;                   \
  }
  if (mt16) { MLD_T32_NS(16) } else { MLD_T32_NS(32) }
#undef MLD_T32_NS
#undef MLD_T32
  count(c);
  check_launch(c, "gemm_tile32");
}

ASrc plain_src(const float* base, int ld) {
  ASrc s;
  s.base = base; s.ld = ld;
  return s;
}
ASrc combine_src(const float* slabs, int nsplit, long long pstride, const float* bias, const float* res,
                 const float* gamma, const float* beta, float* out) {
  ASrc s;
  s.base = slabs; s.ld = 256; s.nsplit = nsplit; s.pstride = pstride; s.bias = bias; s.res = res; s.ldres = 256;
  s.gamma = gamma; s.beta = beta; s.out = out; s.ldout = 256;
  return s;
}

// One chain's slice of the denoiser workspace: rows [row0, row0 + 3R) of every row-indexed buffer.
struct DenView {
  float *X0, *QKV, *FF, *H1, *Ha, *Po, *Pf, *Ps, *S[8], *lat;
  int R;            // samples in this chain's CFG batch (uncond half first)
};

DenView den_view(E* e, int row0, int b0, int R) {
  DenView v;
  const size_t D = e->cfg.latent_dim, F = e->cfg.ff_size, r0 = (size_t)row0;
  v.X0 = e->X0 + r0 * D; v.QKV = e->QKV + r0 * 3 * D; v.FF = e->FF + r0 * F; v.H1 = e->H1 + r0 * D; v.Ha = e->Ha + r0 * D;
  v.Po = e->Po + r0 * D; v.Pf = e->Pf + r0 * D; v.Ps = e->Ps + r0 * D;
  for (int i = 0; i < 8; ++i) v.S[i] = e->S[i] ? e->S[i] + r0 * D : nullptr;
  v.lat = e->lat + (size_t)b0 * D;
  v.R = R;
  return v;
}
long long den_slab(const E* e) { return (long long)6 * e->cfg.max_batch * 256; }

// QKV projection; `x` describes how the layer input rows are obtained (and where they are written back).
void den_qkv(Ctx& c, const DenView& v, const EncLayerP& L, const ASrc& x) {
  Tile32Args a;
  a.src[0] = x; a.nz0 = 1; a.W = L.in_w; a.ldw = 256; a.bias = L.in_b; a.Y = v.QKV; a.ldy = 768; a.M = 3 * v.R; a.N = 768;
  tile32(c, a, 1);
}
// out-projection of the 3-token self-attention (computed while the A tile is assembled) -> raw slab Po
void den_outproj(Ctx& c, const DenView& v, const EncLayerP& L) {
  Tile32Args a;
  a.src[0].base = v.QKV; a.src[0].attn_R = v.R;
  a.nz0 = 1; a.W = L.out_w; a.ldw = 256; a.P = v.Po; a.pstride = 0; a.M = 3 * v.R; a.N = 256;
  tile32(c, a, 1);
}
// h1 = LN1(x + out_proj) assembled on load (written to H1), FF = gelu(h1 W1^T + b1)
void den_ffn1(Ctx& c, const DenView& v, const EncLayerP& L, const float* xn) {
  const int F = c.e->cfg.ff_size;
  Tile32Args a;
  a.src[0] = combine_src(v.Po, 1, 0, L.out_b, xn, L.n1_w, L.n1_b, v.H1);
  a.nz0 = 1; a.W = L.l1_w; a.ldw = 256; a.bias = L.l1_b; a.act = 1; a.Y = v.FF; a.ldy = F; a.M = 3 * v.R; a.N = F;
  tile32(c, a, 1);
}
// FFN2 as ff_size/256 K-slices -> raw slabs Pf; bias, residual and norm2 are applied by whoever reads them
void den_ffn2(Ctx& c, const DenView& v, const EncLayerP& L) {
  const int F = c.e->cfg.ff_size;
  Tile32Args a;
  a.src[0] = plain_src(v.FF, F);
  a.nz0 = F / 256; a.W = L.l2_w; a.ldw = F; a.P = v.Pf; a.pstride = den_slab(c.e); a.M = 3 * v.R; a.N = 256;
  tile32(c, a, F / 256);
}
int den_ffn_slabs(const E* e) { return e->fused_ffn ? e->cfg.ff_size / kFfnHS : e->cfg.ff_size / 256; }
ASrc den_layer_output(E* e, const DenView& v, const EncLayerP& L, float* write_back) {   // LN2(sum Pf + b2 + h1)
  return combine_src(v.Pf, den_ffn_slabs(e), den_slab(e), L.l2_b, v.H1, L.n2_w, L.n2_b, write_back);
}
// linear1 + GELU + linear2 in one launch (kernels/fused_layer.hpp): h1 = LN1(x + out_proj) assembled on load (written
// to H1), raw FFN2 partial slabs -> Pf[ff_size/128]
void den_ffn_fused(Ctx& c, const DenView& v, const EncLayerP& L, const float* xn) {
  FfnFusedArgs a;
  a.src = combine_src(v.Po, 1, 0, L.out_b, xn, L.n1_w, L.n1_b, v.H1);
  a.W1 = L.l1_w; a.b1 = L.l1_b; a.W2 = L.l2_w; a.P = v.Pf; a.pstride = den_slab(c.e); a.M = 3 * v.R; a.F = c.e->cfg.ff_size;
  dim3 grid((a.M + 15) / 16, a.F / kFfnHS);
  MLD_LAUNCH((den_ffn_fused_kernel<1>), grid, dim3(512), kFfnLdsBytes, c.stream, a);
  count(c);
  check_launch(c, "den_ffn_fused");
}

// SkipTransformerEncoder over the 3-token sequences (cross_attention.py:41-64).  Leaves the last layer's
// FFN2 slabs in Pf and its norm1 output in H1; the caller applies norm2 + encoder.norm (FinalArgs).
void denoiser_body(Ctx& c, const DenView& v) {
  E* e = c.e;
  const int nb = (e->cfg.num_layers - 1) / 2, L = e->cfg.num_layers;
  ASrc x = plain_src(v.X0, 256);
  const float* xn = v.X0;                  // where the (normalised) layer input lives, for the norm1 residual
  for (int l = 0; l < L; ++l) {
    const EncLayerP& P_ = e->den[l];
    den_qkv(c, v, P_, x);
    den_outproj(c, v, P_);
    if (e->fused_ffn) {
      den_ffn_fused(c, v, P_, xn);
    } else {
      den_ffn1(c, v, P_, xn);
      den_ffn2(c, v, P_);
    }
    if (l + 1 == L) break;
    if (l < nb) {
      // next layer input = LN2(...), kept in S[l] for the skip connection (written by the next QKV prologue)
      x = den_layer_output(e, v, P_, v.S[l]);
      xn = v.S[l];
    } else {
      // Linear(cat[x, skip]) as two K slices (cross_attention.py:56-58): slice 0 assembles x = LN2(...) on load,
      // slice 1 reads the stored skip activation; the sum + bias is assembled by the next QKV prologue.
      const int i = l - nb;
      Tile32Args a;
      a.src[0] = den_layer_output(e, v, P_, nullptr);
      a.src[1] = plain_src(v.S[nb - 1 - i], 256);
      a.nz0 = 1;
      a.W = P(e, "denoiser.encoder.linear_blocks." + std::to_string(i) + ".weight"); a.ldw = 512;
      a.P = v.Ps; a.pstride = den_slab(e); a.M = 3 * v.R; a.N = 256;
      tile32(c, a, 2);
      x = combine_src(v.Ps, 2, den_slab(e), P(e, "denoiser.encoder.linear_blocks." + std::to_string(i) + ".bias"), nullptr, nullptr,
                      nullptr, v.Ha);
      xn = v.Ha;
    }
  }
}

FinalArgs den_final_args(E* e, const DenView& v) {
  const EncLayerP& L = e->den.back();
  FinalArgs f;
  f.P = v.Pf; f.nsplit = den_ffn_slabs(e); f.pstride = den_slab(e);
  f.b2 = L.l2_b; f.H1 = v.H1; f.g2 = L.n2_w; f.be2 = L.n2_b;
  f.gf = P(e, "denoiser.encoder.norm.weight"); f.bef = P(e, "denoiser.encoder.norm.bias");
  return f;
}

// emb_proj = Sequential(ReLU, Linear) (mld_denoiser.py:65-68) for `rows` text rows -> dst[rows][D]; the
// bias already holds + pe[2] (token 2 of the sequence).
void text_projection(Ctx& c, const float* text_emb, int rows, float* dst) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, TD = e->cfg.text_dim;
  GemmArgs g = lin_args(text_emb, TD, TD, P(e, "denoiser.emb_proj.1.weight"), e->text_bias, dst, D, rows, D);
  g.relu_in = 1;
  gemm(c, g);
}

// time-MLP rows for `n` timestep embeddings already in `temb0` -> out[n, D] (+pe[1] folded in the bias)
void time_mlp(Ctx& c, const float* temb0, float* mid, float* out, int n) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, TD = time_width(e);
  GemmArgs a = lin_args(temb0, TD, TD, P(e, "denoiser.time_embedding.linear_1.weight"),
                        P(e, "denoiser.time_embedding.linear_1.bias"), mid, D, n, D);
  a.act = ACT_SILU;
  gemm(c, a);
  gemm(c, lin_args(mid, D, D, P(e, "denoiser.time_embedding.linear_2.weight"), e->time_b2pe, out, D, n, D));
}

// One decoder layer over M = B*T frame rows with memory = the sample's latent (cross_attention.py:323-345).
int pick_nkt(int T) { return T <= 64 ? 4 : T <= 112 ? 7 : T <= 208 ? 13 : 18; }

void dec_attention(Ctx& c, int B, int T, const int32_t* lens = nullptr) {
  if (!lens) lens = c.e->lens_dev;
  E* e = c.e;
  const int H = e->cfg.num_heads;
  const int nkt = pick_nkt(T);
  const size_t shmem = (size_t)2 * nkt * 16 * 68 * sizeof(float);
  dim3 grid(B * H), block(512);
  switch (nkt) {
    case 4: MLD_LAUNCH((attn_decode_kernel<4>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H); break;
    case 7: MLD_LAUNCH((attn_decode_kernel<7>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H); break;
    case 13: MLD_LAUNCH((attn_decode_kernel<13>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H); break;
    default: MLD_LAUNCH((attn_decode_kernel<18>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H); break;
  }
  count(c);
  check_launch(c, "attn_decode");
}

void dec_layer(Ctx& c, int l, const float* xin, float* xout, int B, int T) {
  E* e = c.e;
  const DecLayerP& L = e->dec[l];
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size, M = B * T;
  gemm(c, lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, M, 3 * D));
  dec_attention(c, B, T);
  // out-proj + residual + norm1, then the 1-key cross-attention (a per-sample vector) + norm2
  GemmArgs o = lin_args(e->AO, D, D, L.out_w, L.out_b, e->H1, D, M, D);
  o.res = xin; o.ldres = D; o.g1 = L.n1_w; o.b1 = L.n1_b;
  o.cvec = e->cvec + (size_t)l * e->cfg.max_batch * D; o.ldcvec = D; o.rows_per_group = T;
  o.g2 = L.n2_w; o.b2 = L.n2_b;
  gemm_ln(c, o);
  GemmArgs f1 = lin_args(e->H1, D, D, L.l1_w, L.l1_b, e->FF, F, M, F);
  f1.act = ACT_GELU;
  gemm(c, f1);
  GemmArgs f2 = lin_args(e->FF, F, F, L.l2_w, L.l2_b, xout, D, M, D);
  f2.res = e->H1; f2.ldres = D; f2.g1 = L.n3_w; f2.b1 = L.n3_b;
  gemm_ln(c, f2);
}

void skip_linear(Ctx& c, const std::string& prefix, int i, const float* x, const float* skip, float* y, int M) {
  E* e = c.e;
  const int D = e->cfg.latent_dim;
  GemmArgs g;
  g.A = x; g.lda = D; g.K1 = D; g.A2 = skip; g.lda2 = D; g.K2 = D;
  g.W = P(e, prefix + ".linear_blocks." + std::to_string(i) + ".weight"); g.ldw = 2 * D;
  g.bias = P(e, prefix + ".linear_blocks." + std::to_string(i) + ".bias");
  g.Y = y; g.ldy = D; g.M = M; g.N = D;
  gemm(c, g);
}

// MldVae.decode (mld_vae.py:186-248).  z [B, D]; lens_dev already holds the lengths.
void decode_body(Ctx& c, const float* z, int B, int T, float* feats_out) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, NF = e->cfg.nfeats, nb = (e->cfg.num_layers - 1) / 2, M = B * T;
  const int L = vae_layers(e);
  // cross-attention with ONE memory token: softmax == 1, so the sub-layer adds
  // out_proj(v_proj(z_b)) to every frame of sample b (exact).  All layers at once.
  {
    GemmArgs v = lin_args(z, D, D, e->dec[0].cin_w + (size_t)2 * D * D, e->dec[0].cin_b + 2 * D, e->cv1, D, B, D);
    v.sW = (long long)e->dec_layer_stride; v.sBias = (long long)e->dec_layer_stride; v.sY = (long long)e->cfg.max_batch * D;
    gemm(c, v, L);
    GemmArgs o = lin_args(e->cv1, D, D, e->dec[0].cout_w, e->dec[0].cout_b, e->cvec, D, B, D);
    o.sA = (long long)e->cfg.max_batch * D; o.sW = (long long)e->dec_layer_stride; o.sBias = (long long)e->dec_layer_stride;
    o.sY = (long long)e->cfg.max_batch * D;
    gemm(c, o, L);
  }
  {
    // time queries = zeros + PE rows (learned: mld_vae.py:216-222; sinusoidal: actor_vae.py:221-222)
    MLD_LAUNCH(init_queries_kernel, dim3(std::min(2048, (M * D / 4 + 255) / 256)), dim3(256), 0, c.stream, e->X0,
               P(e, is_actor(e) ? "vae.decoder.sequence_pos_encoding.pe" : "vae.query_pos_decoder.pe"), B, T, D);
    count(c);
    check_launch(c, "init_queries");
  }
  if (is_actor(e)) {
    // ActorAgnosticDecoder (actor_vae.py:224-235): plain stack, no skip links, no final LayerNorm
    const float* xin = e->X0;
    for (int l = 0; l < L; ++l) {
      float* xout = (l & 1) ? e->Hb : e->Ha;
      dec_layer(c, l, xin, xout, B, T);
      xin = xout;
    }
    GemmArgs f = lin_args(xin, D, D, P(e, "vae.decoder.final_layer.weight"), P(e, "vae.decoder.final_layer.bias"), feats_out, NF, M, NF);
    f.lens = e->lens_dev; f.rows_per_group = T;   // output[~mask.T] = 0 (actor_vae.py:231)
    gemm(c, f);
    return;
  }
  const float* x = e->X0;
  for (int l = 0; l < nb; ++l) {
    dec_layer(c, l, x, e->S[l], B, T);
    x = e->S[l];
  }
  dec_layer(c, nb, x, e->Ha, B, T);
  for (int i = 0; i < nb; ++i) {
    skip_linear(c, "vae.decoder", i, e->Ha, e->S[nb - 1 - i], e->Hb, M);
    dec_layer(c, nb + 1 + i, e->Hb, e->Ha, B, T);
  }
  MLD_LAUNCH(layernorm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, c.stream, (const float*)e->Ha, e->LNO,
             P(e, "vae.decoder.norm.weight"), P(e, "vae.decoder.norm.bias"), M);
  count(c);
  check_launch(c, "layernorm_rows");
  GemmArgs f = lin_args(e->LNO, D, D, P(e, "vae.final_layer.weight"), P(e, "vae.final_layer.bias"), feats_out, NF, M, NF);
  f.lens = e->lens_dev; f.rows_per_group = T;   // output[~mask.T] = 0 (mld_vae.py:245)
  gemm(c, f);
}


// One post-norm encoder layer over M = B*S token rows with a key-padding mask (cross_attention.py:259-272),
// on the decoder's kernels: packed in-proj GEMM, masked MFMA attention, out-proj + res + norm1, FFN.
void venc_layer(Ctx& c, const EncLayerP& L, const float* xin, float* xout, int B, int S) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size, M = B * S;
  gemm(c, lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, M, 3 * D));
  dec_attention(c, B, S, e->lens2_dev);
  GemmArgs o = lin_args(e->AO, D, D, L.out_w, L.out_b, e->H1, D, M, D);
  o.res = xin; o.ldres = D; o.g1 = L.n1_w; o.b1 = L.n1_b;
  gemm_ln(c, o);
  GemmArgs f1 = lin_args(e->H1, D, D, L.l1_w, L.l1_b, e->FF, F, M, F);
  f1.act = ACT_GELU;
  gemm(c, f1);
  GemmArgs f2 = lin_args(e->FF, F, F, L.l2_w, L.l2_b, xout, D, M, D);
  f2.res = e->H1; f2.ldres = D; f2.g1 = L.n2_w; f2.b1 = L.n2_b;
  gemm_ln(c, f2);
}

// MldVae.encode (mld_vae.py:124-184): feats [B,T,nfeats] -> mu, logvar (and latent = mu + exp(logvar)^0.5 * eps).
void encode_body(Ctx& c, const float* feats, int B, int T, const float* eps, float* latent, float* mu, float* logvar) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, NF = e->cfg.nfeats, KP = (NF + 31) / 32 * 32, nb = (e->cfg.num_layers - 1) / 2;
  const int S = T + 2, M = B * S;
  // skel_embedding: K = 263 is padded to 288 so the MFMA K chunks stay full (zeros contribute nothing)
  MLD_LAUNCH(pad_cols_kernel, dim3(std::min(4096, (B * T * KP + 255) / 256)), dim3(256), 0, c.stream, feats, e->FF, B * T, NF, KP);
  count(c);
  check_launch(c, "pad_cols");
  const bool actor = is_actor(e);
  {
    GemmArgs g = lin_args(e->FF, KP, KP, e->WskelP, P(e, actor ? "vae.encoder.skel_embedding.bias" : "vae.skel_embedding.bias"), e->LNO, D,
                          B * T, D);
    gemm(c, g);
  }
  // [token 0, token 1, frames] + positional rows (MldVae: global_motion_token + learned PE, mld_vae.py:150-163;
  // ActorVae: [mu_token, logvar_token] + sinusoidal PE, actor_vae.py:141-163)
  MLD_LAUNCH(enc_tokens_kernel, dim3(std::min(4096, (M * D / 4 + 255) / 256)), dim3(256), 0, c.stream, (const float*)e->LNO,
             P(e, actor ? "vae.encoder.mu_token" : "vae.global_motion_token"),
             P(e, actor ? "vae.encoder.sequence_pos_encoding.pe" : "vae.query_pos_encoder.pe"), e->X0, B, T, D);
  count(c);
  check_launch(c, "enc_tokens");
  if (actor) {
    // ActorAgnosticEncoder (actor_vae.py:164-170): stock nn.TransformerEncoder, no skip links, NO final norm
    const float* xin = e->X0;
    for (int l = 0; l < (int)e->venc.size(); ++l) {
      float* xout = (l & 1) ? e->Hb : e->Ha;
      venc_layer(c, e->venc[l], xin, xout, B, S);
      xin = xout;
    }
    MLD_LAUNCH(enc_finish_kernel, dim3(B), dim3(256), 0, c.stream, xin, (const float*)nullptr, (const float*)nullptr, eps, latent, mu,
               logvar, S);
    count(c);
    check_launch(c, "enc_finish");
    return;
  }
  const float* x = e->X0;
  for (int l = 0; l < nb; ++l) {
    venc_layer(c, e->venc[l], x, e->S[l], B, S);
    x = e->S[l];
  }
  venc_layer(c, e->venc[nb], x, e->Ha, B, S);
  for (int i = 0; i < nb; ++i) {
    skip_linear(c, "vae.encoder", i, e->Ha, e->S[nb - 1 - i], e->Hb, M);
    venc_layer(c, e->venc[nb + 1 + i], e->Hb, e->Ha, B, S);
  }
  MLD_LAUNCH(enc_finish_kernel, dim3(B), dim3(256), 0, c.stream, (const float*)e->Ha, P(e, "vae.encoder.norm.weight"),
             P(e, "vae.encoder.norm.bias"), eps, latent, mu, logvar, S);
  count(c);
  check_launch(c, "enc_finish");
}

void joints_body(Ctx& c, const float* feats, int B, int T, float* joints) {
  E* e = c.e;
  if (T <= 256) {
    MLD_LAUNCH((feats2joints_kernel<256>), dim3(B), dim3(256), 0, c.stream, feats, joints, P(e, "mean"), P(e, "std"), T,
               e->cfg.nfeats, e->cfg.njoints);
  } else {
    MLD_LAUNCH((feats2joints_kernel<512>), dim3(B), dim3(256), 0, c.stream, feats, joints, P(e, "mean"), P(e, "std"), T,
               e->cfg.nfeats, e->cfg.njoints);
  }
  count(c);
  check_launch(c, "feats2joints");
}

// ------------------------------------------------------------------------------------------------------------------
// Diffusion-only variant (BASELINE config 4): trans_dec denoiser on raw motion, d = 512 (kernels/novae.hpp).
// Row layout: sample-major rows r*T + t of the CFG batch (r < R = 2B), 512 floats per row.
void novae_ln(Ctx& c, const float* x, const float* res, const float* g, const float* b, float* y, int M) {
  MLD_LAUNCH((add_layernorm_rows_kernel<512>), dim3((M + 3) / 4), dim3(256), 0, c.stream, x, res, g, b, y, M);
  count(c);
  check_launch(c, "add_layernorm_rows");
}

void novae_self_attention(Ctx& c, int R, int T) {
  E* e = c.e;
  const int H = e->cfg.num_heads, nkt = pick_nkt(T), nqt = (T + 15) / 16;
  const size_t shmem = (size_t)nkt * 16 * 132 * sizeof(float);
  dim3 grid(R * H, (nqt + 7) / 8), block(512);
  const int* nolens = nullptr;    // the reference passes no key-padding mask to the trans_dec denoiser (mld_denoiser.py:215)
  switch (nkt) {
    case 4: MLD_LAUNCH((attn_seq_kernel<4, 128>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
    case 7: MLD_LAUNCH((attn_seq_kernel<7, 128>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
    case 13: MLD_LAUNCH((attn_seq_kernel<13, 128>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
    default: MLD_LAUNCH((attn_seq_kernel<18, 128>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
  }
  count(c);
  check_launch(c, "attn_seq");
}

// K|V of the memory tokens for all layers at once (blockIdx.z = layer): dst[l][rows][2D] = src · Wkv_l^T + bkv_l
void novae_memory_kv(Ctx& c, const float* src, int rows, float* dst, long long dst_layer_stride) {
  E* e = c.e;
  const int D = e->cfg.latent_dim;
  GemmArgs g = lin_args(src, D, D, e->ndec[0].cin_w + (size_t)D * D, e->ndec[0].cin_b + D, dst, 2 * D, rows, 2 * D);
  g.sW = (long long)e->ndec_layer_stride; g.sBias = (long long)e->ndec_layer_stride; g.sY = dst_layer_stride;
  gemm(c, g, e->cfg.num_layers);
}

// MldDenoiser.forward, trans_dec branch, for the M = R*T rows whose zero-padded features are in e->FF [M][KP].
// tkv: K|V of the time token, layer l at tkv + l*tkv_stride; text-token K|V in e->XKV [L][2*max_batch][2D].
void novae_denoiser_body(Ctx& c, int R, int T, const float* tkv, long long tkv_stride, float* eps_out) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size, NF = e->cfg.nfeats, KP = novae_kp(e), M = R * T;
  gemm(c, lin_args(e->FF, KP, KP, e->WskelP, P(e, "denoiser.pose_embd.bias"), e->X0, D, M, D));
  MLD_LAUNCH(add_pe_mod_kernel, dim3(std::min(4096, (M * (D / 4) + 255) / 256)), dim3(256), 0, c.stream, e->X0,
             P(e, "denoiser.query_pos.pe"), (long long)M, T, D);
  count(c);
  check_launch(c, "add_pe_mod");
  for (int l = 0; l < e->cfg.num_layers && !c.rc; ++l) {
    const DecLayerP& L = e->ndec[l];
    gemm(c, lin_args(e->X0, D, D, L.in_w, L.in_b, e->QKV, 3 * D, M, 3 * D));
    novae_self_attention(c, R, T);
    gemm(c, lin_args(e->AO, D, D, L.out_w, L.out_b, e->Ha, D, M, D));
    novae_ln(c, e->Ha, e->X0, L.n1_w, L.n1_b, e->H1, M);
    gemm(c, lin_args(e->H1, D, D, L.cin_w, L.cin_b, e->Hb, D, M, D));                    // cross-attention queries
    MLD_LAUNCH((cross2_kernel<512, 128>), dim3((M + 3) / 4), dim3(256), 0, c.stream, (const float*)e->Hb, tkv + (size_t)l * tkv_stride,
               (const float*)(e->XKV + (size_t)l * 2 * e->cfg.max_batch * 2 * D), e->AO, M, T);
    count(c);
    check_launch(c, "cross2");
    gemm(c, lin_args(e->AO, D, D, L.cout_w, L.cout_b, e->Ha, D, M, D));
    novae_ln(c, e->Ha, e->H1, L.n2_w, L.n2_b, e->X0, M);
    GemmArgs f1 = lin_args(e->X0, D, D, L.l1_w, L.l1_b, e->FF, F, M, F);
    f1.act = ACT_GELU;
    gemm(c, f1);
    gemm(c, lin_args(e->FF, F, F, L.l2_w, L.l2_b, e->Ha, D, M, D));
    novae_ln(c, e->Ha, e->X0, L.n3_w, L.n3_b, e->X0, M);       // in place: a wave reads its whole row before writing it
  }
  novae_ln(c, e->X0, nullptr, P(e, "denoiser.decoder.norm.weight"), P(e, "denoiser.decoder.norm.bias"), e->H1, M);
  GemmArgs f = lin_args(e->H1, D, D, P(e, "denoiser.pose_proj.weight"), P(e, "denoiser.pose_proj.bias"), eps_out, NF, M, NF);
  f.lens = e->lens_dev; f.rows_per_group = T;                   // sample[~mask.T] = 0 (mld_denoiser.py:219-221)
  gemm(c, f);
}

void novae_pad_input(Ctx& c, const float* x, long long rows, int dup) {
  E* e = c.e;
  const int KP = novae_kp(e);
  MLD_LAUNCH(dup_pad_rows_kernel, dim3((unsigned)std::min<long long>(8192, (rows * KP + 255) / 256)), dim3(256), 0, c.stream, x, e->FF, rows,
             e->cfg.nfeats, KP, dup);
  count(c);
  check_launch(c, "dup_pad_rows");
}

// text token of the memory: emb_proj(text) + mem_pos.pe[1] -> TP [rows][D], then its K|V for every layer -> XKV
void novae_text_memory(Ctx& c, const float* text, int rows) {
  E* e = c.e;
  text_projection(c, text, rows, e->TP);
  novae_memory_kv(c, e->TP, rows, e->XKV, (long long)2 * e->cfg.max_batch * 2 * e->cfg.latent_dim);
}

// MLD.forward after the text encoder with vae_type 'no' (mld.py:232-242,264,290-360).  lens_dev holds lengths ++ lengths.
int enqueue_sample_novae(E* e, hipStream_t stream, const float* text, const float* init_lat, int B, int T, const float* step_noise,
                         unsigned long long seed, float* feats_out, float* joints_out) {
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, NF = e->cfg.nfeats, n = e->cfg.num_inference_steps;
  const long long nel = (long long)B * T * NF;
  const float guidance = e->cfg.guidance_scale > 1.0f ? e->cfg.guidance_scale : 1.0f;
  e->launches[0] = e->launches[1] = e->launches[2] = 0;
  e->phase = 0;
  HIP_TRY(e, hipMemcpyAsync(e->lat, init_lat, nel * sizeof(float), hipMemcpyDeviceToDevice, stream));   // init_noise_sigma = 1
  novae_text_memory(c, text, 2 * B);
  for (int s = 0; s < n && !c.rc; ++s) {
    novae_pad_input(c, e->lat, (long long)B * T, 2);                                      // torch.cat([latents] * 2)
    novae_denoiser_body(c, 2 * B, T, e->TKV + (size_t)s * 2 * D, (long long)n * 2 * D, e->feats_int);
    MLD_LAUNCH(cfg_ddpm_step_kernel, dim3((unsigned)std::min<long long>(4096, (nel / 4 + 255) / 256)), dim3(256), 0, stream,
               (const float*)e->feats_int, (const float*)(e->feats_int + nel), (const float*)e->lat,
               step_noise ? step_noise + (size_t)s * nel : (const float*)nullptr, e->lat, nel, guidance,
               ddpm_coef(e, e->timesteps[s]), seed, (unsigned)s);
    count(c);
    check_launch(c, "cfg_ddpm_step");
  }
  if (c.rc) return c.rc;
  if (feats_out) HIP_TRY(e, hipMemcpyAsync(feats_out, e->lat, nel * sizeof(float), hipMemcpyDeviceToDevice, stream));   // "decode" = identity (mld.py:241-242)
  if (joints_out) {
    e->phase = 2;
    joints_body(c, e->lat, B, T, joints_out);
  }
  return c.rc;
}

// Everything mld.py:232-240,264 does after the text encoder.  The reverse loop is latency bound (a few
// hundred rows per launch), and samples never interact, so the batch is cut into `nchains` sub-batches
// whose 50-step chains run on parallel branches (side streams forked from / joined to `stream`; inside
// a capture they become parallel branches of the hipGraph).  The MFMA-bound decode runs on the whole batch.
// rows of token 2 for an action CFG batch of R rows -> dst[R][D] (labels already in labels_dev)
void action_rows(Ctx& c, int R, int nuncond, float* dst) {
  E* e = c.e;
  MLD_LAUNCH(action_rows_kernel, dim3(R), dim3(256), 0, c.stream, dst, P(e, "denoiser.emb_proj.action_embedding"),
             P(e, "denoiser.query_pos.pe") + 2 * e->cfg.latent_dim, (const int*)e->labels_dev, nuncond);
  count(c);
  check_launch(c, "action_rows");
}

// `text` == nullptr selects the action condition (labels_dev holds the 2B labels).
int enqueue_sample(E* e, hipStream_t stream, const float* text, const float* init_lat, int B, int T,
                   float* lat_out, float* feats_out, float* joints_out) {
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, n = e->cfg.num_inference_steps;
  // guidance_scale <= 1: the reference runs the conditional batch alone (mld.py:300,316-340); u + 1*(c-u) is that batch
  const float guidance = e->cfg.guidance_scale > 1.0f ? e->cfg.guidance_scale : 1.0f;
  e->launches[0] = e->launches[1] = e->launches[2] = 0;
  e->phase = 0;
  if (text) text_projection(c, text, 2 * B, e->TP);
  else action_rows(c, 2 * B, B, e->TP);
  int nch = std::min(e->nchains, B);
  const int Bc = (B + nch - 1) / nch;
  nch = (B + Bc - 1) / Bc;
#if !defined(MLDHIP_SIM)
  if (nch > 1) {
    hipError_t s = hipEventRecord(e->ev_fork, stream);
    for (int ch = 1; ch < nch && s == hipSuccess; ++ch) s = hipStreamWaitEvent(e->side[ch - 1], e->ev_fork, 0);
    if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "fork: %s", hipGetErrorString(s));
  }
#endif
  int rc = 0;
  for (int ch = 0; ch < nch; ++ch) {
    const int b0 = ch * Bc, bc = std::min(Bc, B - b0);
#if !defined(MLDHIP_SIM)
    Ctx cc{e, ch == 0 ? stream : e->side[ch - 1]};
#else
    Ctx cc{e, stream};
#endif
    const DenView v = den_view(e, 6 * b0, b0, 2 * bc);
    MLD_LAUNCH(init_chain_kernel, dim3(bc), dim3(256), 0, cc.stream, init_lat + (size_t)b0 * D, v.lat, v.X0,
               P(e, "denoiser.query_pos.pe"), (const float*)e->T1, (const float*)e->TP, B, b0, bc, 1.0f /* init_noise_sigma */);
    count(cc);
    check_launch(cc, "init_chain");
    for (int s = 0; s < n && !cc.rc; ++s) {
      denoiser_body(cc, v);
      const float* t1n = (s + 1 < n) ? e->T1 + (size_t)(s + 1) * D : nullptr;
      MLD_LAUNCH(den_final_step_kernel, dim3(bc), dim3(256), 0, cc.stream, den_final_args(e, v), v.lat, v.X0,
                 P(e, "denoiser.query_pos.pe"), t1n, bc, guidance, ddim_coef(e, e->timesteps[s]));
      count(cc);
      check_launch(cc, "den_final_step");
    }
    if (cc.rc && !rc) rc = cc.rc;
#if !defined(MLDHIP_SIM)
    if (ch > 0) {   // join (also on error paths, so a capture can always be closed)
      hipError_t s = hipEventRecord(e->ev_join[ch - 1], e->side[ch - 1]);
      if (s == hipSuccess) s = hipStreamWaitEvent(stream, e->ev_join[ch - 1], 0);
      if (s != hipSuccess && !rc) rc = e->fail(MLDHIP_EHIP, "join: %s", hipGetErrorString(s));
    }
#endif
  }
  if (rc) return rc;
  if (lat_out) {
    hipError_t s = hipMemcpyAsync(lat_out, e->lat, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream);
    if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "latents copy: %s", hipGetErrorString(s));
  }
  if (feats_out || joints_out) {
    e->phase = 1;
    float* f = feats_out ? feats_out : e->feats_int;
    decode_body(c, e->lat, B, T, f);
    if (joints_out) {
      e->phase = 2;
      joints_body(c, f, B, T, joints_out);
    }
  }
  return c.rc;
}

int validate_lengths(E* e, const int32_t* lengths, int B, int* Tmax) {
  if (!lengths) return e->fail(MLDHIP_EINVAL, "lengths_host is NULL");
  if (B < 1 || B > e->cfg.max_batch) return e->fail(MLDHIP_EINVAL, "batch %d outside [1, max_batch=%d]", B, e->cfg.max_batch);
  int t = 0;
  for (int i = 0; i < B; ++i) {
    if (lengths[i] < 1 || lengths[i] > e->cfg.max_frames)
      return e->fail(MLDHIP_EINVAL, "lengths[%d]=%d outside [1, max_frames=%d]", i, lengths[i], e->cfg.max_frames);
    t = std::max(t, lengths[i]);
  }
  *Tmax = t;
  return 0;
}

}  // namespace

// ======================================================================================= C ABI

extern "C" {

int mldhip_abi_version(void) { return MLDHIP_ABI_VERSION; }

void mldhip_default_config(mldhip_config* c) {
  std::memset(c, 0, sizeof *c);
  c->struct_size = sizeof(mldhip_config);
  c->latent_dim = 256; c->latent_size = 1; c->ff_size = 1024; c->num_layers = 9; c->num_heads = 4;
  c->nfeats = 263; c->njoints = 22; c->text_dim = 768; c->max_batch = 64; c->max_frames = 196;
  c->num_train_timesteps = 1000; c->num_inference_steps = 50; c->steps_offset = 1; c->set_alpha_to_one = 0;
  c->beta_start = 0.00085f; c->beta_end = 0.012f; c->guidance_scale = 7.5f;
  c->precision = MLDHIP_PREC_F32; c->use_graph = 1;
  c->condition = MLDHIP_COND_TEXT; c->nclasses = 0; c->vae_arch = MLDHIP_VAE_MLD; c->vae_num_layers = 0;
  c->denoiser_arch = MLDHIP_ARCH_TRANS_ENC; c->scheduler_type = MLDHIP_SCHED_DDIM;
  c->max_in_flight = 1;
}

const char* mldhip_last_error(mldhip_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int mldhip_create(const mldhip_config* cfg, int device, mldhip_handle** out) {
  auto bad = [&](const char* m) { g_last_error = m; return MLDHIP_EINVAL; };
  if (!cfg || !out) return bad("null argument");
  if (cfg->struct_size != (int32_t)sizeof(mldhip_config)) return bad("mldhip_config.struct_size mismatch (ABI skew)");
  const bool novae = cfg->vae_arch == MLDHIP_VAE_NONE;
  if (cfg->vae_arch != MLDHIP_VAE_MLD && cfg->vae_arch != MLDHIP_VAE_ACTOR && !novae) return bad("vae_arch must be mld, actor or none");
  if (cfg->denoiser_arch != MLDHIP_ARCH_TRANS_ENC && cfg->denoiser_arch != MLDHIP_ARCH_TRANS_DEC) return bad("denoiser_arch must be trans_enc or trans_dec");
  if (cfg->scheduler_type != MLDHIP_SCHED_DDIM && cfg->scheduler_type != MLDHIP_SCHED_DDPM) return bad("scheduler_type must be ddim or ddpm");
  if (novae != (cfg->denoiser_arch == MLDHIP_ARCH_TRANS_DEC) || novae != (cfg->scheduler_type == MLDHIP_SCHED_DDPM))
    return bad("supported combinations: (vae mld|actor, trans_enc, ddim) as in config_mld_*.yaml, or (vae none, trans_dec, ddpm) as in config_novae_humanml3d.yaml");
  if (novae) {
    if (cfg->latent_dim != 512 || cfg->latent_size != 1 || cfg->num_heads * 128 != 512) return bad("diffusion-only variant: latent_dim [1, 512], 4 heads of 128");
    if (cfg->condition != MLDHIP_COND_TEXT) return bad("diffusion-only variant: text condition only");
    if (cfg->num_layers < 1 || cfg->num_layers > 24) return bad("num_layers must be 1..24");
  } else {
    if (cfg->latent_dim != 256 || cfg->latent_size != 1) return bad("latent models: latent_dim [1, 256] only");
    if (cfg->num_heads * 64 != cfg->latent_dim) return bad("head_dim must be 64");
    if (cfg->num_layers < 3 || cfg->num_layers % 2 == 0 || cfg->num_layers > 17) return bad("num_layers must be odd, 3..17 (SkipTransformer)");
  }
  if ((cfg->ff_size != 256 && cfg->ff_size != 512 && cfg->ff_size != 1024) || cfg->text_dim % 32) return bad("ff_size must be 256, 512 or 1024 and text_dim % 32 == 0");
  if (cfg->max_batch < 1 || cfg->max_frames < 1 || cfg->max_frames > 288) return bad("max_batch >= 1, 1 <= max_frames <= 288");
  if (cfg->condition != MLDHIP_COND_TEXT && cfg->condition != MLDHIP_COND_ACTION) return bad("condition must be text or action");
  if (cfg->condition == MLDHIP_COND_ACTION && (cfg->nclasses < 1 || cfg->nclasses > 4096)) return bad("action condition needs 1 <= nclasses <= 4096");
  if (cfg->vae_num_layers < 0 || cfg->vae_num_layers > 17) return bad("vae_num_layers must be 0..17");
  if (cfg->vae_arch != MLDHIP_VAE_ACTOR && (cfg->nfeats < 67 || cfg->njoints != 22)) return bad("HumanML3D layout expected: nfeats >= 67, njoints 22");
  if (cfg->nfeats < 1 || cfg->nfeats > 1024) return bad("nfeats must be 1..1024");
  if (cfg->num_inference_steps < 1 || cfg->num_train_timesteps % cfg->num_inference_steps) return bad("num_train_timesteps must be a multiple of num_inference_steps");
  if (cfg->scheduler_type == MLDHIP_SCHED_DDIM &&
      (cfg->num_inference_steps - 1) * (cfg->num_train_timesteps / cfg->num_inference_steps) + cfg->steps_offset >= cfg->num_train_timesteps)
    return bad("steps_offset pushes the first timestep past num_train_timesteps");
  if (cfg->precision != MLDHIP_PREC_F32 && cfg->precision != MLDHIP_PREC_BF16X3_DECODE) return bad("unsupported precision");
  if (cfg->max_in_flight < 1 || cfg->max_in_flight > 8) return bad("max_in_flight must be 1..8");
#if !defined(MLDHIP_SIM)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_last_error = "no HIP device visible (libmldhip has no CPU path)"; return MLDHIP_ENODEV; }
  if (device < 0 || device >= ndev) return bad("device index out of range");
  if (hipSetDevice(device) != hipSuccess) { g_last_error = "hipSetDevice failed"; return MLDHIP_EHIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_last_error = std::string("libmldhip is built for gfx950 only; device is ") + prop.gcnArchName;
    return MLDHIP_ENODEV;
  }
#endif
  auto* e = new mldhip_engine();
  const char* m_gemm = std::getenv("MLDHIP_GEMM");        // process-wide A/B knobs, re-read at every create
  g_staged_gemm = !(m_gemm && std::strcmp(m_gemm, "direct") == 0);
  const char* m_g8 = std::getenv("MLDHIP_GEMM8");
  g_gemm8 = !(m_g8 && std::atoi(m_g8) == 0);
  const char* m_small = std::getenv("MLDHIP_SMALL_M");
  g_small_m = m_small ? std::atoi(m_small) : 256;
  if (const char* m = std::getenv("MLDHIP_TILE16")) e->tile16 = std::atoi(m) != 0;
  if (const char* m = std::getenv("MLDHIP_FUSED_FFN")) e->fused_ffn = std::atoi(m) != 0;
  if (const char* m = std::getenv("MLDHIP_T32_KH")) e->t32_kh = std::atoi(m) == 2 ? 2 : 1;
  e->nchains = 1;   // measured: parallel chains do not shorten the sequential depth
  if (const char* m = std::getenv("MLDHIP_CHAINS")) e->nchains = std::max(1, std::min(8, std::atoi(m)));
  e->cfg = *cfg;
  e->device = device;
  declare_params(e);
  build_schedule(e);
  auto fail_create = [&](int code) { g_last_error = e->err; mldhip_destroy(e); return code; };
  if (hipMalloc((void**)&e->arena, e->arena_floats * sizeof(float)) != hipSuccess) { e->err = "hipMalloc(weights) failed"; return fail_create(MLDHIP_EHIP); }
  // ---- workspace carve
  const size_t D = cfg->latent_dim, F = cfg->ff_size, TD = std::max(cfg->text_dim, cfg->latent_dim), NF = cfg->nfeats;
  const size_t Bm = cfg->max_batch, Tm = cfg->max_frames, n = cfg->num_inference_steps, L = cfg->num_layers;
  const size_t Lv = std::max<size_t>(L, vae_layers(e));
  const size_t rows = std::max(Bm * (Tm + 2), 6 * Bm);   // decoder: B*T frame rows; encoder: B*(T+2) token rows
  const size_t KP = (NF + 31) / 32 * 32;                 // feature width padded to the MFMA K chunk
  size_t off = 0;
  auto& carve = e->carve;
  auto want = [&](float** p, size_t nfl) { carve.push_back({p, off}); off += align_up(nfl); };
  if (is_novae(e)) {
    // diffusion-only: M = 2*B*T rows of width 512; raw-motion latents [B][T][NF]; eps of the CFG batch [2B][T][NF]
    const size_t r2 = 2 * Bm * Tm, KPn = novae_kp(e);
    want(&e->X0, r2 * D); want(&e->Ha, r2 * D); want(&e->Hb, r2 * D); want(&e->H1, r2 * D); want(&e->LNO, 0);
    for (int i = 0; i < 8; ++i) want(&e->S[i], 0);
    want(&e->QKV, r2 * 3 * D); want(&e->AO, r2 * D); want(&e->FF, r2 * std::max(F, KPn));
    want(&e->lat, Bm * Tm * NF); want(&e->zbuf, 0);
    want(&e->Po, 0); want(&e->Pf, 0); want(&e->Ps, 0); want(&e->TP, 2 * Bm * D);
    want(&e->T1, n * D); want(&e->temb0, n * TD); want(&e->tmid, n * D);
    want(&e->text_bias, D); want(&e->time_b2pe, D); want(&e->t1_one, D); want(&e->temb0_one, TD + D);
    want(&e->cv1, 0); want(&e->cvec, 0);
    want(&e->WskelP, D * KPn);
    want(&e->feats_int, 2 * Bm * Tm * NF); want(&e->joints_int, Bm * Tm * cfg->njoints * 3);
    want(&e->TKV, L * n * 2 * D); want(&e->XKV, L * 2 * Bm * 2 * D); want(&e->TKV_one, L * 2 * D);
  } else {
  want(&e->X0, rows * D); want(&e->Ha, rows * D); want(&e->Hb, rows * D); want(&e->H1, rows * D); want(&e->LNO, rows * D);
  for (int i = 0; i < 8; ++i) want(&e->S[i], (i < (int)(L - 1) / 2) ? rows * D : 0);
  want(&e->QKV, rows * 3 * D); want(&e->AO, rows * D); want(&e->FF, rows * F);
  want(&e->lat, Bm * D); want(&e->zbuf, Bm * D);
  want(&e->Po, 6 * Bm * D); want(&e->Pf, 8 * 6 * Bm * D); want(&e->Ps, 2 * 6 * Bm * D); want(&e->TP, 2 * Bm * D);
  want(&e->T1, n * D); want(&e->temb0, n * TD); want(&e->tmid, n * D);
  want(&e->text_bias, D); want(&e->time_b2pe, D); want(&e->t1_one, D); want(&e->temb0_one, TD + D);
  want(&e->cv1, Lv * Bm * D); want(&e->cvec, Lv * Bm * D);
  want(&e->WskelP, D * KP);
  want(&e->feats_int, Bm * Tm * NF); want(&e->joints_int, Bm * Tm * cfg->njoints * 3);
  want(&e->text_in, 2 * Bm * TD); want(&e->lat_in, Bm * D);
  }
  e->ws_floats = off;
  e->ctxs.resize(cfg->max_in_flight);
  for (auto& x : e->ctxs) {
    if (hipMalloc((void**)&x.ws, off * sizeof(float)) != hipSuccess) { e->err = "hipMalloc(workspace) failed"; return fail_create(MLDHIP_EHIP); }
    if (hipMemset(x.ws, 0, off * sizeof(float)) != hipSuccess) { e->err = "hipMemset(workspace) failed"; return fail_create(MLDHIP_EHIP); }
    if (hipMalloc((void**)&x.lens, 2 * Bm * sizeof(int32_t)) != hipSuccess || hipMalloc((void**)&x.lens2, Bm * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&x.labels, 2 * Bm * sizeof(int32_t)) != hipSuccess) { e->err = "hipMalloc(lens) failed"; return fail_create(MLDHIP_EHIP); }
#if !defined(MLDHIP_SIM)
    if (hipEventCreateWithFlags(&x.done, hipEventDisableTiming) != hipSuccess) { e->err = "event create failed"; return fail_create(MLDHIP_EHIP); }
#endif
  }
  bind_context(e, 0);
#if !defined(MLDHIP_SIM)
  if (hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) { e->err = "hipStreamCreate failed"; return fail_create(MLDHIP_EHIP); }
  for (int i = 0; i < 7; ++i) {
    if (hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming) != hipSuccess) { e->err = "side stream/event create failed"; return fail_create(MLDHIP_EHIP); }
  }
  if (hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) { e->err = "event create failed"; return fail_create(MLDHIP_EHIP); }
  // the decoder attention keeps K and V of one (sample, head) in LDS: up to 2*18*16*68*4 = 153 KiB
  const int big = 2 * 18 * 16 * 68 * 4;
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  const int big128 = 18 * 16 * 132 * 4;   // attn_seq_kernel<*,128>: one operand (K, then V) of up to 288 keys x 132 floats = 148.5 KiB
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<7, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<13, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<18, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
#define MLD_T32_ATTR(NS)                                                                                                    \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<32, NS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes); \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<32, NS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes);  \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<16, NS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes); \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<16, NS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes);
  MLD_T32_ATTR(0) MLD_T32_ATTR(1) MLD_T32_ATTR(2) MLD_T32_ATTR(4) MLD_T32_ATTR(8)
  (void)hipFuncSetAttribute((const void*)den_ffn_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLdsBytes);
#undef MLD_T32_ATTR
  (void)hipGetLastError();
#endif
  *out = e;
  return MLDHIP_OK;
}

void mldhip_destroy(mldhip_handle* e) {
  if (!e) return;
#if !defined(MLDHIP_SIM)
  for (auto& x : e->ctxs) {
    for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
    if (x.done) (void)hipEventDestroy(x.done);
  }
  if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
  for (int i = 0; i < 7; ++i) { if (e->side[i]) (void)hipStreamDestroy(e->side[i]); if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]); }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
#endif
  if (e->arena) (void)hipFree(e->arena);
  for (auto& x : e->ctxs) {
    if (x.ws) (void)hipFree(x.ws);
    if (x.lens) (void)hipFree(x.lens);
    if (x.lens2) (void)hipFree(x.lens2);
    if (x.labels) (void)hipFree(x.labels);
  }
  if (e->trace_buf) (void)hipFree(e->trace_buf);
  delete e;
}

int mldhip_load_tensor(mldhip_handle* e, const char* key, const void* data, const int64_t* shape, int32_t ndim,
                       int32_t dtype, int32_t src_is_device) {
  if (!e || !key || !data || (!shape && ndim > 0)) return e ? e->fail(MLDHIP_EINVAL, "null argument") : MLDHIP_EINVAL;
  if (dtype != MLDHIP_F32) return e->fail(MLDHIP_EINVAL, "tensor %s: only float32 tensors are accepted", key);
  auto it = e->index.find(key);
  if (it == e->index.end()) {
    static const char* ignorable[] = {
                                      "denoiser.mem_pos.", "text_encoder.", "t2m_", "vae.dist_layer."};
    for (auto p : ignorable)
      if (std::strncmp(key, p, std::strlen(p)) == 0) return 1;
    return e->fail(MLDHIP_EINVAL, "unexpected key %s", key);
  }
  Param& p = e->params[it->second];
  size_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= size_t(shape[i]);
  bool same = (size_t)ndim == p.shape.size();
  for (int i = 0; same && i < ndim; ++i) same = shape[i] == p.shape[i];
  if (!same && numel == p.numel && (ndim == 1 || p.shape.size() == 1)) same = true;   // tolerate squeezed vectors
  if (!same) return e->fail(MLDHIP_EINVAL, "shape mismatch for %s (expected %zu elements, got %zu)", key, p.numel, numel);
  HIP_TRY(e, hipMemcpy(e->arena + p.offset, data, p.numel * sizeof(float), src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  p.loaded = true;
  e->finalized = false;
  return MLDHIP_OK;
}

int mldhip_missing_keys(mldhip_handle* e, char* buf, int64_t buflen) {
  if (!e) return MLDHIP_EINVAL;
  int missing = 0;
  int64_t pos = 0;
  for (auto& p : e->params)
    if (!p.loaded) {
      ++missing;
      if (buf && pos + (int64_t)p.key.size() + 1 < buflen) {
        std::memcpy(buf + pos, p.key.c_str(), p.key.size() + 1);
        pos += p.key.size() + 1;
      }
    }
  if (buf && pos < buflen) buf[pos] = 0;
  return missing;
}

int mldhip_finalize_weights(mldhip_handle* e, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  // A group (denoiser / vae decoder / mean+std) must be loaded completely or not at all; ops of an
  // absent group fail with MLDHIP_ESTATE, sample() needs all three.
  int have[4] = {0, 0, 0, 0}, total[4] = {0, 0, 0, 0};
  for (auto& p : e->params) { total[p.group]++; have[p.group] += p.loaded; }
  for (auto& p : e->params)
    if (!p.loaded && have[p.group] != 0) return e->fail(MLDHIP_ENOKEY, "missing tensor %s (strict load)", p.key.c_str());
  if (have[0] + have[1] + have[2] + have[3] == 0) return e->fail(MLDHIP_ENOKEY, "no tensors loaded");
  for (int g = 0; g < 4; ++g) e->group_ready[g] = total[g] > 0 && have[g] == total[g];
  hipStream_t stream = (hipStream_t)stream_;
  bind_layers(e);
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, TD = time_width(e), n = e->cfg.num_inference_steps;
  HIP_TRY(e, hipDeviceSynchronize());                   // no call may be in flight on any context while tables are rebuilt
  for (int k = 0; k < (int)e->ctxs.size(); ++k) {       // the derived tables live in each context's workspace
  bind_context(e, k);
  if (e->group_ready[0]) {
    // PE-folded biases: token 1 (time) gets pe[1], token 2 (text) gets pe[2] (mld_denoiser.py:187,196)
    // (trans_dec: the memory tokens [time, text] get mem_pos.pe[0], pe[1] instead, mld_denoiser.py:213)
    const float* pe_time = is_novae(e) ? P(e, "denoiser.mem_pos.pe") : P(e, "denoiser.query_pos.pe") + D;
    const float* pe_text = pe_time + D;
    MLD_LAUNCH(add_rows_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, e->time_b2pe, P(e, "denoiser.time_embedding.linear_2.bias"), pe_time, 1, D);
    if (!is_action(e)) MLD_LAUNCH(add_rows_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, e->text_bias, P(e, "denoiser.emb_proj.1.bias"), pe_text, 1, D);
    if (check_launch(c, "add_rows")) return c.rc;
    // time-MLP output for every scheduler timestep (sample independent; embeddings.py:245-305)
    std::vector<float> host((size_t)n * TD);
    for (int s = 0; s < n; ++s) timestep_sincos(float(e->timesteps[s]), TD, host.data() + (size_t)s * TD);
    HIP_TRY(e, hipMemcpy(e->temb0, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    time_mlp(c, e->temb0, e->tmid, e->T1, n);
    if (c.rc) return c.rc;
    if (is_novae(e)) {
      // the time token's K|V for every (layer, scheduler step) depend on weights only; pose_embd.weight padded to KP
      novae_memory_kv(c, e->T1, n, e->TKV, (long long)n * 2 * D);
      const int NF = e->cfg.nfeats, KP = novae_kp(e);
      MLD_LAUNCH(pad_cols_kernel, dim3((D * KP + 255) / 256), dim3(256), 0, stream, P(e, "denoiser.pose_embd.weight"), e->WskelP, D, NF, KP);
      if (check_launch(c, "pad_cols")) return c.rc;
    }
  }
  if (e->group_ready[3]) {
    const int NF = e->cfg.nfeats, KP = (NF + 31) / 32 * 32;
    MLD_LAUNCH(pad_cols_kernel, dim3((D * KP + 255) / 256), dim3(256), 0, stream,
               P(e, is_actor(e) ? "vae.encoder.skel_embedding.weight" : "vae.skel_embedding.weight"), e->WskelP, D, NF, KP);
    if (check_launch(c, "pad_cols")) return c.rc;
  }
  }
  HIP_TRY(e, hipStreamSynchronize(stream));
#if !defined(MLDHIP_SIM)
  for (auto& x : e->ctxs) {
    for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
    x.graphs.clear();
    x.used = false;
  }
#endif
  bind_context(e, 0);
  e->next_ctx = 0;
  e->finalized = true;
  return MLDHIP_OK;
}

}  // extern "C"

namespace {
// shared body of mldhip_sample / mldhip_sample_action (text_emb_dev == nullptr <=> action labels given)
int sample_impl(mldhip_handle* e, const float* text_emb_dev, const int32_t* actions_host, const float* init_latents_dev,
                const int32_t* lengths_host, int32_t B, float* latents_out_dev, float* feats_out_dev, float* joints_out_dev,
                void* stream_) {
  if (!e->finalized) return e->fail(MLDHIP_ESTATE, "mldhip_sample before mldhip_finalize_weights");
  if (!e->group_ready[0] || !e->group_ready[1] || (joints_out_dev && !e->group_ready[2]))
    return e->fail(MLDHIP_ESTATE, "mldhip_sample needs denoiser.*, vae.decoder.* (and mean/std for joints) loaded");
  if (!init_latents_dev) return e->fail(MLDHIP_EINVAL, "null input pointer");
  int T = 0;
  if (int rc = validate_lengths(e, lengths_host, B, &T)) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  if (actions_host) {
    for (int i = 0; i < B; ++i)
      if (actions_host[i] < 0 || actions_host[i] >= e->cfg.nclasses)
        return e->fail(MLDHIP_EINVAL, "actions[%d]=%d outside [0, nclasses=%d)", i, actions_host[i], e->cfg.nclasses);
    // cond = cat(zeros_like(actions), actions) (mld.py:722-725); the first half is never read (null embedding)
    HIP_TRY(e, hipMemsetAsync(e->labels_dev, 0, (size_t)B * sizeof(int32_t), stream));
    HIP_TRY(e, hipMemcpyAsync(e->labels_dev + B, actions_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  }
#if !defined(MLDHIP_SIM)
  if (e->cfg.use_graph) {
    const size_t D = e->cfg.latent_dim, NF = e->cfg.nfeats;
    const bool want_j = joints_out_dev != nullptr, want_f = feats_out_dev != nullptr || want_j;
    if (text_emb_dev)
      HIP_TRY(e, hipMemcpyAsync(e->text_in, text_emb_dev, (size_t)2 * B * e->cfg.text_dim * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(e, hipMemcpyAsync(e->lat_in, init_latents_dev, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    GraphKey key{B, T, want_f, want_j};
    auto& graphs = e->ctxs[e->cur_ctx].graphs;
    auto it = graphs.find(key);
    if (it == graphs.end()) {
      if (graphs.size() >= 16) {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
      }
      hipGraph_t graph = nullptr;
      HIP_TRY(e, hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
      int rc = enqueue_sample(e, e->cap_stream, text_emb_dev ? e->text_in : nullptr, e->lat_in, B, T, nullptr,
                              want_f ? e->feats_int : nullptr, want_j ? e->joints_int : nullptr);
      hipError_t s = hipStreamEndCapture(e->cap_stream, &graph);
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(s));
      hipGraphExec_t exec = nullptr;
      s = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(s));
      it = graphs.emplace(key, exec).first;
    }
    HIP_TRY(e, hipGraphLaunch(it->second, stream));
    if (latents_out_dev) HIP_TRY(e, hipMemcpyAsync(latents_out_dev, e->lat, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (feats_out_dev) HIP_TRY(e, hipMemcpyAsync(feats_out_dev, e->feats_int, (size_t)B * T * NF * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (joints_out_dev)
      HIP_TRY(e, hipMemcpyAsync(joints_out_dev, e->joints_int, (size_t)B * T * e->cfg.njoints * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return MLDHIP_OK;
  }
#endif
  return enqueue_sample(e, stream, text_emb_dev, init_latents_dev, B, T, latents_out_dev, feats_out_dev, joints_out_dev);
}
}  // namespace

extern "C" {

int mldhip_sample(mldhip_handle* e, const float* text_emb_dev, const float* init_latents_dev, const int32_t* lengths_host,
                  int32_t B, float* latents_out_dev, float* feats_out_dev, float* joints_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the action condition: use mldhip_sample_action");
  if (is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was created for the diffusion-only variant: use mldhip_sample_novae");
  if (joints_out_dev && is_actor(e)) return e->fail(MLDHIP_ESTATE, "joints of the ActorVae feature layout need SMPL (out of scope)");
  if (!text_emb_dev) return e->fail(MLDHIP_EINVAL, "null input pointer");
  return sample_impl(e, text_emb_dev, nullptr, init_latents_dev, lengths_host, B, latents_out_dev, feats_out_dev, joints_out_dev, stream_);
}

int mldhip_sample_action(mldhip_handle* e, const int32_t* actions_host, const float* init_latents_dev, const int32_t* lengths_host,
                         int32_t B, float* latents_out_dev, float* feats_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the text condition: use mldhip_sample");
  if (!actions_host) return e->fail(MLDHIP_EINVAL, "null input pointer");
  return sample_impl(e, nullptr, actions_host, init_latents_dev, lengths_host, B, latents_out_dev, feats_out_dev, nullptr, stream_);
}

}  // extern "C"

namespace {
int denoiser_forward_impl(mldhip_handle* e, const float* sample_dev, int32_t timestep, const float* text_emb_dev,
                          const int32_t* actions_host, int32_t R, float* out_dev, void* stream_) {
  if (!e->finalized || !e->group_ready[0]) return e->fail(MLDHIP_ESTATE, "denoiser_forward before finalize / denoiser.* not loaded");
  if (!sample_dev || !out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  if (R < 1 || R > 2 * e->cfg.max_batch) return e->fail(MLDHIP_EINVAL, "R=%d outside [1, 2*max_batch]", R);
  if (timestep < 0 || timestep >= e->cfg.num_train_timesteps) return e->fail(MLDHIP_EINVAL, "timestep %d out of range", timestep);
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, TD = time_width(e);
  e->phase = 0;
  if (actions_host) {
    for (int i = 0; i < R; ++i)
      if (actions_host[i] < 0 || actions_host[i] >= e->cfg.nclasses)
        return e->fail(MLDHIP_EINVAL, "actions[%d]=%d outside [0, nclasses=%d)", i, actions_host[i], e->cfg.nclasses);
    HIP_TRY(e, hipMemcpyAsync(e->labels_dev, actions_host, (size_t)R * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  }
  std::vector<float> host(TD);
  timestep_sincos(float(timestep), TD, host.data());
  HIP_TRY(e, hipMemcpyAsync(e->temb0_one, host.data(), TD * sizeof(float), hipMemcpyHostToDevice, stream));
  HIP_TRY(e, hipStreamSynchronize(stream));   // `host` is a stack temporary
  time_mlp(c, e->temb0_one, e->temb0_one + TD, e->t1_one, 1);
  const DenView v = den_view(e, 0, 0, R);
  if (text_emb_dev) text_projection(c, text_emb_dev, R, e->X0 + (size_t)2 * R * D);
  else action_rows(c, R, e->cfg.guidance_scale > 1.0f ? R / 2 : 0, e->X0 + (size_t)2 * R * D);   // mld_denoiser.py:253-257
  // token 0 rows: sample + pe[0]; token 1 rows: the time-MLP row (pe[1] already folded in)
  MLD_LAUNCH(add_rows_kernel, dim3((R * D + 255) / 256), dim3(256), 0, stream, e->X0, sample_dev, P(e, "denoiser.query_pos.pe"), R, D);
  MLD_LAUNCH(bcast_rows_kernel, dim3((R * D + 255) / 256), dim3(256), 0, stream, e->X0 + (size_t)R * D, (const float*)e->t1_one, R, D);
  check_launch(c, "assemble");
  denoiser_body(c, v);
  MLD_LAUNCH(den_final_rows_kernel, dim3(R), dim3(256), 0, stream, den_final_args(e, v), out_dev);
  check_launch(c, "final_norm");
  return c.rc;
}
}  // namespace

extern "C" {

int mldhip_denoiser_forward(mldhip_handle* e, const float* sample_dev, int32_t timestep, const float* text_emb_dev,
                            int32_t R, float* out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the action condition: use mldhip_denoiser_forward_action");
  if (is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was created for the diffusion-only variant: use mldhip_denoiser_forward_novae");
  if (!text_emb_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  return denoiser_forward_impl(e, sample_dev, timestep, text_emb_dev, nullptr, R, out_dev, stream_);
}

int mldhip_denoiser_forward_action(mldhip_handle* e, const float* sample_dev, int32_t timestep, const int32_t* actions_host,
                                   int32_t R, float* out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the text condition: use mldhip_denoiser_forward");
  if (!actions_host) return e->fail(MLDHIP_EINVAL, "null pointer");
  return denoiser_forward_impl(e, sample_dev, timestep, nullptr, actions_host, R, out_dev, stream_);
}

int mldhip_sample_novae(mldhip_handle* e, const float* text_emb_dev, const float* init_latents_dev, const int32_t* lengths_host,
                        int32_t B, const float* step_noise_dev, uint64_t seed, float* feats_out_dev, float* joints_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was not created for the diffusion-only variant (vae_arch = MLDHIP_VAE_NONE)");
  if (!e->finalized || !e->group_ready[0] || (joints_out_dev && !e->group_ready[2]))
    return e->fail(MLDHIP_ESTATE, "mldhip_sample_novae needs finalize and denoiser.* (and mean/std for joints) loaded");
  if (!text_emb_dev || !init_latents_dev) return e->fail(MLDHIP_EINVAL, "null input pointer");
  int T = 0;
  if (int rc = validate_lengths(e, lengths_host, B, &T)) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev + B, lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));   // lengths * 2 (mld.py:327-328)
  // ~114 launches of 0.1-2 ms each per step: the GPU, not the host, is the bottleneck -> plain stream launches, no graph
  return enqueue_sample_novae(e, stream, text_emb_dev, init_latents_dev, B, T, step_noise_dev, seed, feats_out_dev, joints_out_dev);
}

int mldhip_denoiser_forward_novae(mldhip_handle* e, const float* sample_dev, int32_t timestep, const float* text_emb_dev,
                                  const int32_t* lengths_host, int32_t R, int32_t T, float* out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was not created for the diffusion-only variant (vae_arch = MLDHIP_VAE_NONE)");
  if (!e->finalized || !e->group_ready[0]) return e->fail(MLDHIP_ESTATE, "denoiser_forward_novae before finalize / denoiser.* not loaded");
  if (!sample_dev || !text_emb_dev || !lengths_host || !out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  if (R < 1 || R > 2 * e->cfg.max_batch) return e->fail(MLDHIP_EINVAL, "R=%d outside [1, 2*max_batch]", R);
  if (T < 1 || T > e->cfg.max_frames) return e->fail(MLDHIP_EINVAL, "T=%d outside [1, max_frames=%d]", T, e->cfg.max_frames);
  if (timestep < 0 || timestep >= e->cfg.num_train_timesteps) return e->fail(MLDHIP_EINVAL, "timestep %d out of range", timestep);
  for (int i = 0; i < R; ++i)
    if (lengths_host[i] < 0 || lengths_host[i] > T) return e->fail(MLDHIP_EINVAL, "lengths[%d]=%d outside [0, T=%d]", i, lengths_host[i], T);
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, TD = time_width(e);
  e->phase = 0;
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lengths_host, (size_t)R * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  std::vector<float> host(TD);
  timestep_sincos(float(timestep), TD, host.data());
  HIP_TRY(e, hipMemcpyAsync(e->temb0_one, host.data(), TD * sizeof(float), hipMemcpyHostToDevice, stream));
  HIP_TRY(e, hipStreamSynchronize(stream));   // `host` is a stack temporary
  time_mlp(c, e->temb0_one, e->temb0_one + TD, e->t1_one, 1);
  novae_memory_kv(c, e->t1_one, 1, e->TKV_one, (long long)2 * D);
  novae_text_memory(c, text_emb_dev, R);
  novae_pad_input(c, sample_dev, (long long)R * T, 1);
  novae_denoiser_body(c, R, T, e->TKV_one, (long long)2 * D, out_dev);
  return c.rc;
}

int mldhip_ddpm_step(mldhip_handle* e, const float* eps_dev, int32_t timestep, const float* sample_dev, const float* noise_dev,
                     uint64_t seed, int32_t step_index, float* prev_dev, int64_t n, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!is_ddpm(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the DDIM scheduler: use mldhip_ddim_step");
  if (!eps_dev || !sample_dev || !prev_dev || n < 1) return e->fail(MLDHIP_EINVAL, "null pointer / n < 1");
  if (timestep < 0 || timestep >= e->cfg.num_train_timesteps) return e->fail(MLDHIP_EINVAL, "timestep %d out of range", timestep);
  Ctx c{e, (hipStream_t)stream_};
  MLD_LAUNCH(cfg_ddpm_step_kernel, dim3((unsigned)std::min<long long>(4096, (n / 4 + 256) / 256)), dim3(256), 0, c.stream, eps_dev,
             (const float*)nullptr, sample_dev, noise_dev, prev_dev, (long long)n, 1.0f, ddpm_coef(e, timestep), (unsigned long long)seed,
             (unsigned)step_index);
  return check_launch(c, "ddpm_step");
}

int mldhip_philox_normal(mldhip_handle* e, float* out_dev, int64_t n, uint64_t seed, int32_t step_index, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!out_dev || n < 1) return e->fail(MLDHIP_EINVAL, "null pointer / n < 1");
  Ctx c{e, (hipStream_t)stream_};
  MLD_LAUNCH(philox_normal_kernel, dim3((unsigned)std::min<long long>(4096, (n / 4 + 256) / 256)), dim3(256), 0, c.stream, out_dev, (long long)n,
             (unsigned long long)seed, (unsigned)step_index);
  return check_launch(c, "philox_normal");
}

int mldhip_vae_decode(mldhip_handle* e, const float* z_dev, const int32_t* lengths_host, int32_t B, float* feats_out_dev,
                      void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!e->finalized || !e->group_ready[1]) return e->fail(MLDHIP_ESTATE, "vae_decode before finalize / vae.* not loaded");
  if (!z_dev || !feats_out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  int T = 0;
  if (int rc = validate_lengths(e, lengths_host, B, &T)) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  Ctx c{e, stream};
  e->phase = 1;
  decode_body(c, z_dev, B, T, feats_out_dev);
  return c.rc;
}

int mldhip_vae_encode(mldhip_handle* e, const float* feats_dev, const int32_t* lengths_host, int32_t B, int32_t T,
                      const float* eps_dev, float* latent_out_dev, float* mu_out_dev, float* logvar_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!e->finalized || !e->group_ready[3]) return e->fail(MLDHIP_ESTATE, "vae_encode before finalize / vae.encoder.* not loaded");
  if (!feats_dev || !mu_out_dev || !logvar_out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  if (eps_dev && !latent_out_dev) return e->fail(MLDHIP_EINVAL, "eps given but latent_out is NULL");
  int Tm = 0;
  if (int rc = validate_lengths(e, lengths_host, B, &Tm)) return rc;
  if (T < Tm || T > e->cfg.max_frames || T + 2 > 288) return e->fail(MLDHIP_EINVAL, "T=%d must satisfy max(lengths) <= T <= min(max_frames, 286)", T);
  hipStream_t stream = (hipStream_t)stream_;
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  e->lens2_host.assign(lengths_host, lengths_host + B);
  for (auto& v : e->lens2_host) v += 2;                       // the two distribution tokens are always attended to
  HIP_TRY(e, hipMemcpyAsync(e->lens2_dev, e->lens2_host.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  Ctx c{e, stream};
  e->phase = 1;
  encode_body(c, feats_dev, B, T, eps_dev, latent_out_dev, mu_out_dev, logvar_out_dev);
  return c.rc;
}

__global__ void ddim_step_kernel(const float* eps, const float* x, float* out, long long n, DdimCoef c) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x0 = (x[i] - c.sqrt_1mat * eps[i]) / c.sqrt_at;
    out[i] = c.sqrt_ap * x0 + c.sqrt_1map * eps[i];
  }
}

int mldhip_ddim_step(mldhip_handle* e, const float* eps_dev, int32_t timestep, const float* sample_dev, float* prev_dev,
                     int64_t n, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (!eps_dev || !sample_dev || !prev_dev || n < 0) return e->fail(MLDHIP_EINVAL, "bad argument");
  if (timestep < 0 || timestep >= e->cfg.num_train_timesteps) return e->fail(MLDHIP_EINVAL, "timestep %d out of range", timestep);
  if (n == 0) return MLDHIP_OK;
  Ctx c{e, (hipStream_t)stream_};
  MLD_LAUNCH(ddim_step_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, c.stream, eps_dev, sample_dev,
             prev_dev, (long long)n, ddim_coef(e, timestep));
  return check_launch(c, "ddim_step");
}

int mldhip_feats2joints(mldhip_handle* e, const float* feats_dev, int32_t B, int32_t T, float* joints_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  if (is_actor(e) || e->cfg.nfeats < 67) return e->fail(MLDHIP_ESTATE, "feats2joints implements the HumanML3D layout only (SMPL-based layouts are out of scope)");
  if (!e->finalized || !e->group_ready[2]) return e->fail(MLDHIP_ESTATE, "feats2joints before finalize / mean,std not loaded");
  if (!feats_dev || !joints_out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  if (B < 1 || T < 1 || T > 512) return e->fail(MLDHIP_EINVAL, "B >= 1 and 1 <= T <= 512 required");
  Ctx c{e, (hipStream_t)stream_};
  e->phase = 2;
  joints_body(c, feats_dev, B, T, joints_out_dev);
  return c.rc;
}


int mldhip_profile_kernel(mldhip_handle* e, const char* name, int32_t B, int32_t T, int32_t iters, double* flops_per_launch,
                          void* stream_) {
  // Launches ONE kernel of the sampling path `iters` times back-to-back on `stream` at its production
  // shape, on the engine's own buffers (call after a sample() so they hold real activations).  The
  // caller brackets the call with events on the same stream (bench.py does) -- no timing happens here.
  if (!e || !name || !flops_per_launch) return MLDHIP_EINVAL;
  if (!e->finalized || !e->group_ready[0] || !e->group_ready[1]) return e->fail(MLDHIP_ESTATE, "profile before finalize");
  if (B < 1 || B > e->cfg.max_batch || T < 1 || T > e->cfg.max_frames || iters < 1) return e->fail(MLDHIP_EINVAL, "bad B/T/iters");
  CtxUse use(e, (hipStream_t)stream_);
  if (use.rc) return use.rc;
  Ctx c{e, (hipStream_t)stream_};
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size, H = e->cfg.num_heads;
  const std::string n = name;
  const bool dec = n.rfind("dec_", 0) == 0;
  const int R = 2 * B;
  const long long M = dec ? (long long)B * T : 3LL * R;
  const int mid = (e->cfg.num_layers - 1) / 2;
  int saved_phase = e->phase;
  e->phase = dec ? 1 : 0;
  const EncLayerP& DL = e->den[mid];
  const DenView v = den_view(e, 0, 0, R);
  for (int it = 0; it < iters && !c.rc; ++it) {
    if (n == "den_qkv") {            // with the LN2-on-load prologue of a typical layer (sums the 4 FFN2 slabs)
      den_qkv(c, v, DL, den_layer_output(e, v, e->den[mid - 1], v.S[mid - 1]));
      *flops_per_launch = 2.0 * M * D * 3 * D;
    } else if (n == "den_outproj") {
      den_outproj(c, v, DL);
      *flops_per_launch = 2.0 * M * D * D + 4.0 * M * 3 * D;
    } else if (n == "den_ffn") {      // linear1 + GELU + linear2 fused (kernels/fused_layer.hpp)
      den_ffn_fused(c, v, DL, v.S[mid - 1]);
      *flops_per_launch = 4.0 * M * D * F;
    } else if (n == "den_ffn1") {
      den_ffn1(c, v, DL, v.S[mid - 1]);
      *flops_per_launch = 2.0 * M * D * F;
    } else if (n == "den_ffn2") {
      den_ffn2(c, v, DL);
      *flops_per_launch = 2.0 * M * D * F;
    } else if (n == "den_final") {
      MLD_LAUNCH(den_final_step_kernel, dim3(B), dim3(256), 0, c.stream, den_final_args(e, v), e->zbuf, e->LNO,
                 P(e, "denoiser.query_pos.pe"), (const float*)e->T1, B, e->cfg.guidance_scale, ddim_coef(e, e->timesteps[0]));
      check_launch(c, "den_final_step");
      *flops_per_launch = 0.0;
    } else if (n == "dec_qkv") {
      gemm(c, lin_args(e->S[0], D, D, e->dec[mid].in_w, e->dec[mid].in_b, e->QKV, 3 * D, (int)M, 3 * D));
      *flops_per_launch = 2.0 * M * D * 3 * D;
    } else if (n == "dec_ffn1") {
      GemmArgs f1 = lin_args(e->H1, D, D, e->dec[mid].l1_w, e->dec[mid].l1_b, e->FF, F, (int)M, F);
      f1.act = ACT_GELU;
      gemm(c, f1);
      *flops_per_launch = 2.0 * M * D * F;
    } else if (n == "dec_ffn2_ln") {
      GemmArgs f2 = lin_args(e->FF, F, F, e->dec[mid].l2_w, e->dec[mid].l2_b, e->Hb, D, (int)M, D);
      f2.res = e->H1; f2.ldres = D; f2.g1 = e->dec[mid].n3_w; f2.b1 = e->dec[mid].n3_b;
      gemm_ln(c, f2);
      *flops_per_launch = 2.0 * M * D * F;
    } else if (n == "dec_outproj_ln") {
      GemmArgs o = lin_args(e->AO, D, D, e->dec[mid].out_w, e->dec[mid].out_b, e->Hb, D, (int)M, D);
      o.res = e->S[0]; o.ldres = D; o.g1 = e->dec[mid].n1_w; o.b1 = e->dec[mid].n1_b;
      o.cvec = e->cvec; o.ldcvec = D; o.rows_per_group = T; o.g2 = e->dec[mid].n2_w; o.b2 = e->dec[mid].n2_b;
      gemm_ln(c, o);
      *flops_per_launch = 2.0 * M * D * D;
    } else if (n == "dec_attn") {
      dec_attention(c, B, T);
      *flops_per_launch = 4.0 * B * H * (double)T * T * 64;
    } else {
      e->phase = saved_phase;
      return e->fail(MLDHIP_EINVAL, "unknown kernel name %s", name);
    }
  }
  e->phase = saved_phase;
  return c.rc;
}

int mldhip_profile_trace(mldhip_handle* e, const char* name, int32_t B, int32_t T, uint64_t* out_host, int64_t cap_u64, void* stream_) {
  // Runs ONE traced launch of a den_* kernel (after 3 untraced warm-ups) and copies back 8 timestamps per
  // wave: [0] start [1] loads landed + prologue [2] LDS written [3] barrier passed [4] MFMAs done
  // [5] stores drained (shader clock), [6]/[7] start/end on the 100 MHz realtime counter.
  if (!e || !name || !out_host) return MLDHIP_EINVAL;
  constexpr int64_t kMax = 512 * 8 * 8;
  if (!e->trace_buf && hipMalloc((void**)&e->trace_buf, kMax * sizeof(uint64_t)) != hipSuccess) return e->fail(MLDHIP_EHIP, "hipMalloc(trace)");
  double fl = 0;
  if (int rc = mldhip_profile_kernel(e, name, B, T, 3, &fl, stream_)) return rc;
  HIP_TRY(e, hipMemsetAsync(e->trace_buf, 0, kMax * sizeof(uint64_t), (hipStream_t)stream_));
  e->trace_on = e->trace_buf;
  int rc = mldhip_profile_kernel(e, name, B, T, 1, &fl, stream_);
  e->trace_on = nullptr;
  if (rc) return rc;
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_));
  const int64_t n = std::min<int64_t>(cap_u64, kMax);
  HIP_TRY(e, hipMemcpy(out_host, e->trace_buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return (int)(n / 64);
}

int mldhip_get_timesteps(mldhip_handle* e, int32_t* out, int32_t n) {
  if (!e || !out) return MLDHIP_EINVAL;
  int m = std::min<int>(n, (int)e->timesteps.size());
  std::memcpy(out, e->timesteps.data(), m * sizeof(int32_t));
  return m;
}

int mldhip_get_alphas_cumprod(mldhip_handle* e, float* out, int32_t n) {
  if (!e || !out) return MLDHIP_EINVAL;
  int m = std::min<int>(n, (int)e->alphas_cumprod.size());
  std::memcpy(out, e->alphas_cumprod.data(), m * sizeof(float));
  return m;
}

int mldhip_get_launch_counts(mldhip_handle* e, int32_t* out) {
  if (!e || !out) return MLDHIP_EINVAL;
  for (int i = 0; i < 3; ++i) out[i] = e->launches[i];
  return MLDHIP_OK;
}

}  // extern "C"
