// Weight contract (SURVEY.md App. B key names), layer binding, scheduler tables, workspace-context plumbing.
// Part of libmldhip's single translation unit (included by ../mldhip.hip, in this order: state, params, dispatch,
// path_latent, path_novae).  Internal linkage throughout (anonymous namespace) except the handle type itself.
#pragma once
#include <mutex>

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
// context's previous user (ALWAYS, also with a single context: two calls on different streams must never share a
// workspace concurrently), binds its buffers; on exit records the context's "done" event on the stream.
struct CtxUse {
  E* e;
  hipStream_t stream;
  int rc = 0;
  CtxUse(E* e_, hipStream_t s) : e(e_), stream(s) {
    const int k = int(e->next_ctx++ % e->ctxs.size());
#if !defined(MLDHIP_SIM)
    WsContext& x = e->ctxs[k];
    if (x.used) {
      hipError_t st = hipStreamWaitEvent(stream, x.done, 0);
      if (st != hipSuccess) rc = e->fail(MLDHIP_EHIP, "hipStreamWaitEvent(context): %s", hipGetErrorString(st));
    }
#endif
    bind_context(e, k);
  }
  ~CtxUse() {
#if !defined(MLDHIP_SIM)
    WsContext& x = e->ctxs[e->cur_ctx];
    (void)hipEventRecord(x.done, stream);
    x.used = true;
#endif
  }
};

// Two cluster launches (kernels/loop_cluster.hpp) must never be dispatched side by side: each would keep its resident workgroups spinning on members that the other
// one's workgroups keep off the CUs (2 x 192 workgroups, 256 CUs) until the 200 ms timeout fails both.  One lane per device and process: a call served by the cluster loop waits
// for the previous such call of ANY handle or stream (the whole call: its event sits behind the decode) and leaves its own event behind; the lane's mutex is held while the
// call is enqueued, so wait / record pairs of host threads do not interleave.  Other kernels beside a cluster launch only delay it (they end); other PROCESSES on the
// same GPU are outside the lane's reach (include/mldhip.h "cluster loop").
struct ClusterLane {
#if !defined(MLDHIP_SIM)
  struct Slot { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool used = false; };
  static Slot& slot(int dev) { static Slot s[16]; return s[dev & 15]; }
  Slot* s = nullptr;
  hipStream_t stream;
  ClusterLane(E* e, hipStream_t st, bool on) : stream(st) {
    if (!on) return;
    s = &slot(e->device);
    s->mu.lock();
    if (!s->ev && hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { s->ev = nullptr; return; }
    if (s->used && s->last != stream) (void)hipStreamWaitEvent(stream, s->ev, 0);
  }
  ~ClusterLane() {
    if (!s) return;
    if (s->ev && hipEventRecord(s->ev, stream) == hipSuccess) { s->last = stream; s->used = true; }
    s->mu.unlock();
  }
#else
  ClusterLane(E*, hipStream_t, bool) {}
#endif
  ClusterLane(const ClusterLane&) = delete;
  ClusterLane& operator=(const ClusterLane&) = delete;
};

// Selects the engine's device for the duration of one C-ABI call and restores the caller's current device on exit
// (two engines on different GPUs in one process, or a host that switched devices since mldhip_create).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int want) {
#if !defined(MLDHIP_SIM)
    if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = hipSetDevice(want) == hipSuccess;
#endif
    (void)want;
  }
  ~DeviceGuard() {
#if !defined(MLDHIP_SIM)
    if (switched) (void)hipSetDevice(prev);
#endif
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A captured graph exec may still be running on the context's last stream: wait for the context's completion event
// before destroying it (HIP does not promise deferred destruction of an in-flight exec).
void drain_context(WsContext& x) {
#if !defined(MLDHIP_SIM)
  if (x.used && x.done) (void)hipEventSynchronize(x.done);
#endif
  (void)x;
}

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

}  // namespace
