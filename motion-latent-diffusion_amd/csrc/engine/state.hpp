// Engine state: parameter table entries, per-layer weight views, workspace contexts, the handle struct.
// Part of libmldhip's single translation unit (included by ../mldhip.hip, in this order: state, params, dispatch,
// path_latent, path_novae).  Internal linkage throughout (anonymous namespace) except the handle type itself.
#pragma once

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
  bool dec_only = false;   // the decode half of a call alone (reads the context's latents): the pipelined form of mldhip_sample_many ("many_pipeline")
  int part = 0;            // loop-only graphs of the pipelined form: 1 = what precedes the cluster launch (condition rows, flag clear), 2 = the cluster launch + its non-finite count; 0 = everything
  bool operator<(const GraphKey& o) const { return std::tie(B, T, feats, joints, dec_only, part) < std::tie(o.B, o.T, o.feats, o.joints, o.dec_only, o.part); }
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
  unsigned long long seed_host = 0;   // stable host copy of the Philox seed while it is uploaded to seed_slot
#if !defined(MLDHIP_SIM)
  hipEvent_t done = nullptr;
  hipEvent_t pre_done = nullptr;               // pipelined sample_many: recorded on the engine's prep stream behind a request's input staging + condition rows + flag clear
  hipEvent_t loop_done = nullptr;              // pipelined sample_many: recorded behind the reverse loop on the caller's stream, waited for by the side stream's decode
  std::map<GraphKey, hipGraphExec_t> graphs;   // captured sample() graphs of this workspace, evicted least-recently-used
  std::vector<GraphKey> graph_lru;             // most recent last
  std::map<std::tuple<int, int, int>, hipGraphExec_t> step_graphs;   // diffusion-only variant: (B, Tmax, chunk) -> captured DDPM steps
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
  const float* final_stream = nullptr;   // split modes, MldVae: vae.final_layer.weight zero-padded to 384 rows as a fragment-ordered stream (kernels/final_strip.hpp); inside ffn_streams
  float* ffn_streams = nullptr;   // split modes: linear1 / linear2 of every decoder / encoder layer as fragment-ordered item streams (kernels/ffn_strip.hpp)
  std::map<const float*, const float*> ffn_stream_of;   // linear1.weight (arena pointer) -> its layer's stream
  std::map<const float*, const float*> gemm_stream_of;  // in_proj / out_proj / skip-linear weight -> its stream (kernels/gemm_strip_x3.hpp)
  float* loop_stream = nullptr;   // sample-major loop (kernels/loop_fused.hpp): GEMM weights of the denoiser re-packed in consumption order
  float* loop_stream_x3 = nullptr;   // ... the same items as split-f16 images (the precision mode with split arithmetic)
  float* loop_small = nullptr;    // ... its biases / LayerNorm parameters, packed; then the DDIM coefficients [n][4]
  float* loop_ddim = nullptr;
  int loop_ips = 0;               // weight items per reverse step (0: the variant is not built for this configuration)
  float* cl_stream = nullptr;     // cluster loop (kernels/loop_cluster.hpp): per column group and wave, the split-f16 fragments in consumption order
  unsigned cl_wave_off[96] = {0}; // ... float offset of (column group, wave)'s sequence: 32 words of the 4-group form, 64 of the 8-group form
  unsigned* cl_wave_off_dev = nullptr;   // ... the same 96 words in device memory (kernel arguments stay small)
  int cluster_groups = 0;         // "cluster_groups": 0 = 8 column groups per token up to 64 motions and 4 above, 4 / 8 = forced (8 only up to 64 motions)
  int cluster_mute = -1, cluster_timeout = 0;   // "cluster_inject" (hooks build only): member that never raises its first flag, and the shortened wait bound that goes with it
  bool cluster_stale = false;     // "cluster_stale" (hooks build only): a launch finds a stale epoch in one of its polled words (what the r05 memset-node fault left behind): the entry check must fail the launch
  int cluster_chunk = 128;        // "cluster_chunk" (hooks build only): motions per cluster launch (a multiple of 8, at most 8 x kClMaxClusters = 128)
  bool cluster_lane = true;       // "cluster_lane" (hooks build only): 0 = no ordering between cluster calls of different streams (the starvation it prevents, on purpose)
  bool cluster_graph = true;      // "cluster_graph" (hooks build only): 0 = calls served by the cluster loop are issued eagerly
  int num_cus = 1 << 20;          // CUs of the device (a partitioned or masked device has fewer than 256): a cluster launch needs a CU per workgroup (simulator: no limit)
  unsigned* cl_host_status = nullptr;   // pinned host word the cluster kernel sets next to its sticky status word: read at the start of every sample call (no device synchronisation) -- a timed-out handle leaves the cluster loop by itself
  bool cluster_foreign = false;   // another PROCESS holds the cluster lane of this device (lock file taken at mldhip_create): this handle never launches the cluster loop
  int cluster_failed = 0;         // a cluster launch reported a timeout / a placement it cannot use: the handle stays on the other loop families
  float* arena_x3 = nullptr;  // split-bf16 image of the arena (precision modes with split-bf16 staged GEMMs; built by finalize)
  size_t arena_floats = 0;
  std::vector<EncLayerP> den;      // execution order
  std::vector<DecLayerP> dec;
  std::vector<EncLayerP> venc;     // VAE encoder layers (same layer type as the denoiser's)
  std::vector<DecLayerP> ndec;     // no-VAE variant: denoiser.decoder.layers.* (TransformerDecoder, cross_attention.py:195-233)
  size_t ndec_layer_stride = 0;
  float* seed_slot = nullptr;     // 8 bytes of workspace: the Philox seed of the call whose captured step graphs are running
  float *TKV = nullptr, *XKV = nullptr, *TKV_one = nullptr;   // memory-token K|V per layer: time [L][n][2D], text [L][2*max_batch][2D]
  // ... folded through the cross-attention's query / out projections (kernels/novae.hpp cross_fold_kernel, "cross_fold"): w | u [L][tokens][H][D], c [L][tokens][H]
  float *TKW = nullptr, *TKU = nullptr, *TKC = nullptr, *XKW = nullptr, *XKU = nullptr, *XKC = nullptr, *TKW_one = nullptr, *TKU_one = nullptr, *TKC_one = nullptr;
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
  float* FS = nullptr;   // sample-major loop: parked skip activations [ceil(max_batch / 8)][nb][48][256]
  float *cl_xbuf = nullptr, *cl_park = nullptr, *cl_flags = nullptr;   // cluster loop: exchange regions [clusters][kClXFloats], parked skip rows [workgroups][nb][16][256], flags [clusters][64] + status [16] (words)
  float *Po, *Pf, *Ps;   // denoiser split-K slabs: out-proj [1], FFN2 [4], skip-linear [2], each [6*max_batch][256]
  unsigned long long* trace_buf = nullptr;   // measurement only (mldhip_profile_trace)
  unsigned long long* trace_on = nullptr;    // non-null while a traced launch is being built
  float* WskelP = nullptr;   // skel_embedding.weight padded to [D][KP]
  int32_t* labels_dev = nullptr; // action labels of the CFG batch [2*max_batch] (uncond half first, ignored there)
  int32_t* lens2_dev = nullptr;  // lengths + 2 (encoder key-padding mask incl. the two distribution tokens)
  std::vector<int32_t> lens2_host;
  float *text_in = nullptr, *lat_in = nullptr;   // graph staging of the caller's inputs
  float* TP;             // text projection rows [2*max_batch][256] (+pe[2]), gathered per chain
  // per-handle options (mldhip_set_option)
  int many_pipeline = 0;     // "many_pipeline": 1 = mldhip_sample_many runs its requests ONE AFTER THE OTHER on the single-request path (cluster loop; every request what mldhip_sample gives it, to the bit) with the decode of request k on the engine's side stream, beside the reverse loop of request k + 1 (needs max_in_flight >= 2); 0 = one chain over all motions of the call
  int small_m = 256;         // "gemm_small_m": row count up to which the register-direct tiny-GEMM shape is used
  int loop_kernel = 0;       // "loop_kernel": 0 auto (by rows / motions), 1 latency kernels (tile32.hpp), 2 throughput kernels (strip.hpp), 3 sample-major persistent loop (loop_fused.hpp), 4 cluster loop (loop_cluster.hpp)
#if defined(MLDHIP_SIM)
  int cluster_max_batch = 0;   // (the functional simulator's tests pick the loop family explicitly)
#else
  int cluster_max_batch = 256; // "cluster_max_batch": auto runs the cluster loop (split mode) for calls of up to this many motions (0: never); at most kClMaxCall = 256 (two launches above 128)
#endif
  int cluster_wt = 0;        // "cluster_wt": 0 (default) = a cluster whose twelve members report one XCC id stores its payloads plain (served by the shared L2; -3.5 % per call), any other cluster write-through; 1 = write-through (sc1) always
  int fused_x3 = 1;          // "fused_x3": in the split precision mode the sample-major loop multiplies on split-f16 MFMAs (0: exact fp32 MFMAs)
  int fused_dbg = 0;         // "fused_dbg": 5 = the split-mode loop with its phase counters (same arithmetic, mldhip_profile_trace "den_loop_phases"); 0 = off
  int fused_min_batch = 0;   // "fused_min_batch": auto picks the sample-major loop from this many motions per call up; 0 = by operand format (320 split-f16, 1 280 fp32)
  int strip_min_rows = 768;  // "strip_min_rows": auto switches to the throughput kernels at 6B >= this many token rows
  int flash_attn = 1;        // "flash_attn": split-bf16 frame-level self-attention key-blocked (attention.hpp attn_flash_x3_kernel): 0 never, 1 auto (>= 512 (sample, head) pairs), 2 always
  int ffn_strip = 1;         // "ffn_strip": register-direct decoder kernels (ffn_strip.hpp, gemm_strip_x3.hpp): 0 off, 1 auto strip height, 4 / 6 = 64 / 96 rows always
  int dec_tail = 1;          // "dec_tail": out-projection + norms + feed-forward block of a decoder layer as one launch (chip-filling launches, split modes)
  int dec_l0_once = 1;       // "dec_l0_once": decoder layer 0 projects its input -- the positional rows, the same for every sample -- once per call ([T] rows instead of [B T])
  int dec_half = 0;          // "dec_half": OPT-IN (default 0 = fp32 Q | K | V and split x3 products: gemm_strip_x3.hpp, attention.hpp).  1 / 4 / 6: split mode, decoder self-attention block on half Q | K | V (kernels/dec_half.hpp): in-projection with half activation rows x split weights (2 matrix instructions per product), Q | K | V stored as halves, attention on plain half operands -- kept only where finalize's probe reads it below MLDHIP_PROBE_TOL_HALF on the handle's weights; 2 = without that veto (A/B tools).  Off by default because it is not safe in general: profiles/r06_decoder_precision.json (heavy-tailed weights on O(1) latents: 6.7e-4 .. 8.8e-4 on the joints)
  int tile_x3 = 1;           // "tile_x3": split-f16 mode runs the latency kernels (tile32.hpp) on split-f16 MFMAs too (0: exact fp32)
  int strip_gemm = 1;        // "strip_gemm": split modes, decoder / encoder in-projection, out-projection (+ LayerNorms) and skip linears on the row-strip kernels (kernels/gemm_strip_x3.hpp); 0 = the staged 64 x 128 / 64 x 256 tiles
  int cross_fold = 1;        // "cross_fold": diffusion-only variant: LayerNorm 1 + the two-token cross-attention sub-layer (query GEMM, attention, out-projection GEMM) + LayerNorm 2 of a trans_dec layer as ONE launch on vectors folded from the memory tokens (kernels/novae.hpp cross2_fold_ln_kernel; exact algebra); 0 = the five launches
  int gemm_pipe = 1;         // "gemm_pipe": diffusion-only variant, split modes: the K >= 512 GEMMs on the software-pipelined 128 x 256 tile (kernels/gemm_pipe.hpp): 1 = launches of >= 2 048 rows, 2 = always (tests); 0 = the 64 x 128 staged tile
  int gemm_pipe_min_rows = 2048;   // (not an option) row count from which the big tile is used: below it its 128-row tiles leave most CUs without a workgroup

  // ---- numeric contract of the split-f16 mode (mldhip_numeric_status; include/mldhip.h "Range contract")
#if defined(MLDHIP_SIM)
  int range_probe = 0;       // "range_probe": the functional simulator pays minutes per probe; its tests switch it on where they test it
#else
  int range_probe = 1;       // "range_probe": finalize compares the split-f16 kernels with the exact-fp32 ones on a probe batch and falls back per stage
#endif
  bool probe_first_call = false; // "range_probe" 2: the next text-conditioned mldhip_sample runs the reverse-loop probe on its own batch first
  bool split_loop_ok = true;     // false: the reverse loop runs on exact-fp32 MFMAs although the handle was created in the split mode
  bool split_decode_ok = true;   // false: decoder / encoder / diffusion-only GEMMs and attention run on exact-fp32 MFMAs
  bool dec_half_ok = true;       // false: the probe read the half-Q|K|V form of the decoder's self-attention block above MLDHIP_PROBE_TOL_HALF: the decoder keeps fp32 Q | K | V and x3 products
  float probe_err_decode_half = -1.f;
  float probe_err_loop = -1.f, probe_err_decode = -1.f;   // probe results (max-abs difference / max-abs reference); -1: not probed
  unsigned* nonfinite = nullptr; // device counter: non-finite values seen in the latents / joints a sample call produced (sticky until read)

  int launches[3] = {0, 0, 0};
  int phase = 0;
  int sample_part = 0;       // while a loop-only graph of the pipelined form is captured: which part enqueue_sample issues (GraphKey.part)

#if !defined(MLDHIP_SIM)
  hipStream_t cap_stream = nullptr;
  hipStream_t prep_stream = nullptr;   // "many_pipeline": the inputs, condition rows and flag clear of request k + 1 run here, beside the cluster launch of request k
  hipEvent_t many_start = nullptr;     // ... ordered behind what the caller's stream held when the call came in
  hipStream_t side_stream = nullptr;   // "many_pipeline": decodes run here, at the lowest stream priority (the cluster launch on the caller's stream gets its CUs first)
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
