// libmldhip: engine + C ABI (include/mldhip.h) for the MLD sampling hot path on MI355X (gfx950).
//
// Host-side structure (what the reference leaves to PyTorch/Lightning, rebuilt natively):
//   ParamTable   one HBM arena holding every weight the path reads, laid out per layer in execution
//                order with a uniform per-layer stride (lets one launch cover all 9 layers' tiny
//                cross-attention GEMMs via blockIdx.z);
//   Workspace    one HBM arena for activations, sized from (max_batch, max_frames) at create();
//   Schedule     DDIM tables (float32, as diffusers keeps them) + the time-MLP output for each of
//                the scheduler's timesteps, computed once at finalize (they depend on weights only);
//   sample()     ~2.1k kernel launches captured once per (B, Tmax, requested outputs) and workspace context into a
//                hipGraph and replayed: the 50-step loop has no host work and no host<->device sync.
//
// Source layout: engine/state.hpp (handle, contexts) -> engine/params.hpp (weight contract, schedules) ->
// engine/dispatch.hpp (kernel selection) -> engine/path_latent.hpp / engine/path_novae.hpp (the model paths) -> the C ABI
// below; kernels/*.hpp hold the device code.  One translation unit: hipcc builds it in one pass for gfx950.
//
// Reference call stack being replaced: mld/models/modeltype/mld.py:216-265,290-360.
#include "../../include/mldhip.h"
#if defined(MLDHIP_HOOKS)
#include "../../include/mldhip_hooks.h"
#endif

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#if !defined(MLDHIP_SIM)
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#endif

#include "kernels/attention.hpp"
#include "kernels/elementwise.hpp"
#include "kernels/gemm.hpp"
#include "kernels/gemm_pipe.hpp"
#include "kernels/novae.hpp"
#include "kernels/rt.hpp"
#include "kernels/tile32.hpp"
#include "kernels/strip.hpp"
#include "kernels/loop_fused.hpp"
#include "kernels/loop_cluster.hpp"
#include "kernels/ffn_strip.hpp"
#include "kernels/gemm_strip_x3.hpp"
#include "kernels/final_strip.hpp"
#include "kernels/dec_half.hpp"

using namespace mld;

#include "engine/state.hpp"
#include "engine/params.hpp"
#include "engine/dispatch.hpp"
#include "engine/path_latent.hpp"
#include "engine/path_novae.hpp"


// ======================================================================================= C ABI

namespace { constexpr int kStepChunk = 20; }                // DDPM steps per captured graph (diffusion-only variant)
namespace { constexpr size_t kGraphCacheCapacity = 48; }   // captured graphs kept per workspace context

namespace {
// One PROCESS per device may launch the cluster loop (kernels/loop_cluster.hpp): its launches need their workgroups resident together, ClusterLane orders them inside a
// process, and two processes interleaving such launches on one GPU would each end partly resident -- every wait runs into its 200 ms bound (advisor r5).  The first
// process that creates a handle on a device takes an advisory lock on a per-device file and keeps it until it exits (released by the kernel on any exit); a process that
// finds it taken -- and is not the owner itself or one of its descendants -- runs every call on the other loop families (mldhip_numeric_info.cluster_loop says so).
// One process per GPU -- torch.distributed ranks -- is unaffected; two ranks sharing a GPU are siblings: the second one is foreign.
bool cluster_lane_owned(int device) {
#if !defined(MLDHIP_SIM)
  static std::mutex mu;
  static int state[64] = {0};     // 0 unknown, 1 owned by this process, 2 foreign
  std::lock_guard<std::mutex> lk(mu);
  int& st = state[device & 63];
  if (st) return st == 1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); st = 1; return true; }
  for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.') *c = '_';
  const std::string path = std::string("/tmp/mldhip_cluster_lane_") + bus + ".lock";      // (a fixed directory: processes with different TMPDIRs must meet at one file)
  // O_NOFOLLOW + regular-file check: /tmp is shared, the name is predictable -- never write through somebody's symlink.  Another user's file (created under their umask) may
  // not be writable: flock works on a read-only descriptor too, only the pid note is skipped then.
  int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
  bool writable = fd >= 0;
  if (fd >= 0) (void)fchmod(fd, 0666);                        // (ours if we created it; EPERM otherwise: ignored)
  else fd = open(path.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {      // no lock file possible (read-only /tmp, a symlink in its place): no coordination, behave as before
    if (fd >= 0) close(fd);
    st = 1;
    return true;
  }
  if (flock(fd, LOCK_EX | LOCK_NB) == 0) {                    // fd stays open for the life of the process; the owner's pid goes into the file
    char buf[32];
    const int n = snprintf(buf, sizeof buf, "%ld\n", (long)getpid());
    if (writable && ftruncate(fd, 0) == 0 && pwrite(fd, buf, (size_t)n, 0) == n) {}
    st = 1;
    return true;
  }
  // taken.  One tenant = the owner's process TREE: a second instance of the library inside the owner (the hooks build beside the production one) and the owner's
  // supervised children (bench.py's rocprofv3 child runs while the parent sits idle) are the owner's business; anybody else is foreign
  long owner = -1;
  {
    char buf[32] = {0};
    if (pread(fd, buf, sizeof buf - 1, 0) > 0) owner = strtol(buf, nullptr, 10);
  }
  close(fd);
  long pid = (long)getpid();
  for (int depth = 0; depth < 32 && pid > 1 && owner > 1; ++depth) {
    if (pid == owner) { st = 1; return true; }
    char sp[64];
    snprintf(sp, sizeof sp, "/proc/%ld/stat", pid);
    FILE* f = fopen(sp, "r");
    if (!f) break;
    char line[512] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) break;
    const char* rp = strrchr(line, ')');                      // pid (comm) state ppid ...: comm may hold spaces and parentheses
    long ppid = -1;
    char state = 0;
    if (!rp || sscanf(rp + 1, " %c %ld", &state, &ppid) != 2) break;
    pid = ppid;
  }
  st = 2;
  return false;
#else
  (void)device;
  return true;
#endif
}
}  // namespace

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
    if (cfg->latent_dim != 256 || cfg->latent_size != 1)
      return bad("mldhip_config.latent_size / latent_dim (model.latent_dim in the YAML): only [1, 256] is built; the reference's [N, 256] ablations "
                 "(N = 2, 5, 7, 10: N + 2 denoiser tokens, mld_denoiser.py:171,187; 2N global / N memory tokens, mld_vae.py:150-163) are not");
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
  if (cfg->precision == 3) return bad("precision 3 (MLDHIP_PREC_FP8_DENOISER of ABI <= 4) was retired in ABI 5: it met no tolerance and was slower than MLDHIP_PREC_F16X3 (include/mldhip.h)");
  if (cfg->precision < MLDHIP_PREC_F32 || cfg->precision > MLDHIP_PREC_BF16) return bad("unsupported precision");
  if (cfg->max_in_flight < 1 || cfg->max_in_flight > 8) return bad("max_in_flight must be 1..8");
#if !defined(MLDHIP_SIM)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_last_error = "no HIP device visible (libmldhip has no CPU path)"; return MLDHIP_ENODEV; }
  if (device < 0 || device >= ndev) return bad("device index out of range");
  hipDeviceProp_t prop;
  int num_cus = 0;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      g_last_error = std::string("libmldhip is built for gfx950 only; device is ") + prop.gcnArchName;
      return MLDHIP_ENODEV;
    }
    num_cus = prop.multiProcessorCount;
  }
#endif
  DeviceGuard dg(device);      // allocations below land on `device`; the caller's current device is restored on return
  auto* e = new mldhip_engine();
  e->cfg = *cfg;
  e->device = device;
#if !defined(MLDHIP_SIM)
  e->num_cus = num_cus;
  e->cluster_foreign = !cluster_lane_owned(device);
  if (hipHostMalloc((void**)&e->cl_host_status, sizeof(unsigned), hipHostMallocMapped) == hipSuccess) *e->cl_host_status = 0u;
  else { e->cl_host_status = nullptr; (void)hipGetLastError(); }
#else
  e->cl_host_status = new unsigned(0u);
#endif
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
    {
      const size_t Hn = (size_t)cfg->num_heads;      // folded memory tokens ("cross_fold"): w, u [L][tokens][H][D], c [L][tokens][H]
      want(&e->TKW, L * n * Hn * D); want(&e->TKU, L * n * Hn * D); want(&e->TKC, L * n * Hn);
      want(&e->XKW, L * 2 * Bm * Hn * D); want(&e->XKU, L * 2 * Bm * Hn * D); want(&e->XKC, L * 2 * Bm * Hn);
      want(&e->TKW_one, L * Hn * D); want(&e->TKU_one, L * Hn * D); want(&e->TKC_one, L * Hn);
    }
    want(&e->seed_slot, 2);
  } else {
  want(&e->X0, rows * D); want(&e->Ha, rows * D); want(&e->Hb, rows * D); want(&e->H1, rows * D); want(&e->LNO, rows * D);
  for (int i = 0; i < 8; ++i) want(&e->S[i], (i < (int)(L - 1) / 2) ? rows * D : 0);
  want(&e->QKV, rows * 3 * D); want(&e->AO, rows * D); want(&e->FF, rows * F);
  want(&e->lat, Bm * D); want(&e->zbuf, Bm * D);
  want(&e->FS, (Bm + 7) / 8 * ((L - 1) / 2) * 48 * D);
  {
    // cluster loop (kernels/loop_cluster.hpp): at most kClMaxClusters clusters of 8 motions, 12 workgroups each, launched in rows of 8 XCD slots
    const size_t ncl = D == 256 ? std::min<size_t>(kClMaxClusters, (Bm + 7) / 8) : 0, wgs = std::max<size_t>(8 * kClMembers * ((ncl + 7) / 8), 8 * kClMembersMax);
    want(&e->cl_xbuf, ncl * kClXFloats); want(&e->cl_park, wgs * ((L - 1) / 2) * 16 * 256); want(&e->cl_flags, ncl ? ncl * kClFlagWords + 16 : 0);
  }
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
    if (hipEventCreateWithFlags(&x.done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x.loop_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x.pre_done, hipEventDisableTiming) != hipSuccess) { e->err = "event create failed"; return fail_create(MLDHIP_EHIP); }
#endif
  }
  bind_context(e, 0);
  if (hipMalloc((void**)&e->nonfinite, sizeof(unsigned)) != hipSuccess || hipMemset(e->nonfinite, 0, sizeof(unsigned)) != hipSuccess) {
    e->err = "hipMalloc(non-finite counter) failed";
    return fail_create(MLDHIP_EHIP);
  }
#if !defined(MLDHIP_SIM)
  if (hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) { e->err = "hipStreamCreate failed"; return fail_create(MLDHIP_EHIP); }
  // the decoder attention keeps K and V of one (sample, head) in LDS: up to 2*18*16*68*4 = 153 KiB
  const int big = 2 * 18 * 16 * 68 * 4;
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute((const void*)attn_decode_x3_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_x3_lds_bytes<7>());
  (void)hipFuncSetAttribute((const void*)attn_decode_x3_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_x3_lds_bytes<13>());
  (void)hipFuncSetAttribute((const void*)attn_decode_x3_kernel<18>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_x3_lds_bytes<18>());
  const int big128 = 18 * 16 * 132 * 4;   // attn_seq_kernel<*,128>: one operand (K, then V) of up to 288 keys x 132 floats = 148.5 KiB
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<7, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<13, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_kernel<18, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, big128);
  (void)hipFuncSetAttribute((const void*)attn_seq_x3_kernel<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (attn_seq_x3_lds_bytes<4, 128>()));
  (void)hipFuncSetAttribute((const void*)attn_seq_x3_kernel<7, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (attn_seq_x3_lds_bytes<7, 128>()));
  (void)hipFuncSetAttribute((const void*)attn_seq_x3_kernel<13, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (attn_seq_x3_lds_bytes<13, 128>()));
  (void)hipFuncSetAttribute((const void*)attn_seq_x3_kernel<18, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (attn_seq_x3_lds_bytes<18, 128>()));
  (void)hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<6, 1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_gemm_lds_bytes<6, 1, true>()));
  (void)hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<4, 1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_gemm_lds_bytes<4, 1, true>()));
  (void)hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<4, 2, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_gemm_lds_bytes<4, 2, false>()));
  (void)hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<6, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_gemm_lds_bytes<6, 1, false>()));
  (void)hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<4, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_gemm_lds_bytes<4, 1, false>()));
  (void)hipFuncSetAttribute((const void*)ffn_strip_x3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn_strip_lds_bytes<6>());
  (void)hipFuncSetAttribute((const void*)ffn_strip_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn_strip_lds_bytes<4>());
  (void)hipFuncSetAttribute((const void*)ffn_strip_x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn_strip_lds_bytes<3>());
  (void)hipFuncSetAttribute((const void*)ffn_strip_x3_kernel<3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ffn_strip_lds_bytes<3>());
  (void)hipFuncSetAttribute((const void*)final_strip_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, final_strip_lds_bytes());
  (void)hipFuncSetAttribute((const void*)attn_flash_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFlashLdsBytes);
  (void)hipFuncSetAttribute((const void*)attn_flash_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFlashHLdsBytes);
  (void)hipFuncSetAttribute((const void*)strip_inproj_h_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, inproj_h_lds_bytes<4>());
  (void)hipFuncSetAttribute((const void*)strip_inproj_h_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, inproj_h_lds_bytes<6>());
  (void)hipFuncSetAttribute((const void*)attn_flash128_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFlash128LdsBytes);
  (void)hipFuncSetAttribute((const void*)cross2_fold_ln_kernel<512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, kC2LdsBytes);
  (void)hipFuncSetAttribute((const void*)cross_fold_kernel<512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, kCrossFoldLdsBytes);
  (void)hipFuncSetAttribute((const void*)attn_decode_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_x3_lds_bytes<4>());
  (void)hipFuncSetAttribute((const void*)gemm_pipe_x3_kernel<2, 4, 4, 4, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (gemm_pipe_lds_bytes<2, 4, 4, 4>()));
  (void)hipFuncSetAttribute((const void*)gemm_pipe_x3_kernel<2, 4, 4, 4, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (gemm_pipe_lds_bytes<2, 4, 4, 4>()));
  (void)hipFuncSetAttribute((const void*)den_loop_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes);
  (void)hipFuncSetAttribute((const void*)den_loop_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes);
#if defined(MLDHIP_HOOKS)
  (void)hipFuncSetAttribute((const void*)den_loop_kernel<true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes);
#endif
  (void)hipFuncSetAttribute((const void*)den_cluster_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kClLdsBytes);
  (void)hipFuncSetAttribute((const void*)den_cluster_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kClLdsBytes);
  (void)hipFuncSetAttribute((const void*)den_cluster_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kClLdsBytes);
  (void)hipFuncSetAttribute((const void*)den_cluster_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kClLdsBytes);
#define MLD_T32_ATTR1(MT, NS, TR, PR) \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<MT, NS, TR, PR, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes); \
  (void)hipFuncSetAttribute((const void*)gemm_tile32_kernel<MT, NS, TR, PR, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kT32LdsBytes);
#define MLD_T32_ATTR(NS)                                                                                                    \
  MLD_T32_ATTR1(32, NS, false, PREC_F32) MLD_T32_ATTR1(32, NS, true, PREC_F32) MLD_T32_ATTR1(16, NS, false, PREC_F32) MLD_T32_ATTR1(16, NS, true, PREC_F32) \
  MLD_T32_ATTR1(32, NS, false, PREC_BF16) MLD_T32_ATTR1(16, NS, false, PREC_BF16)                                             \
  MLD_T32_ATTR1(32, NS, false, PREC_BF16X3) MLD_T32_ATTR1(16, NS, false, PREC_BF16X3)                                         \
  MLD_T32_ATTR1(32, NS, true, PREC_BF16X3) MLD_T32_ATTR1(16, NS, true, PREC_BF16X3)
  MLD_T32_ATTR(0) MLD_T32_ATTR(1) MLD_T32_ATTR(2) MLD_T32_ATTR(4)
#undef MLD_T32_ATTR
#undef MLD_T32_ATTR1
#define MLD_STRIP_ATTR8(NS, ACT)                                                                                            \
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<NS, 1, false, PREC_F32, ACT, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<1, 2>())); \
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<NS, 1, false, PREC_BF16, ACT, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<1, 2>()));;
  MLD_STRIP_ATTR8(0, 0) MLD_STRIP_ATTR8(1, 0) MLD_STRIP_ATTR8(1, 1) MLD_STRIP_ATTR8(2, 0)
#undef MLD_STRIP_ATTR8
#define MLD_STRIP_ATTR8S(NS)                                                                                                \
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<NS, 2, false, PREC_F32, 0, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<2, 1>())); \
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<NS, 2, false, PREC_BF16, 0, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<2, 1>()));;
  MLD_STRIP_ATTR8S(1) MLD_STRIP_ATTR8S(2)
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<1, 1, false, PREC_F32, 1, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<1, 2>()));
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<2, 1, false, PREC_F32, 0, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<1, 2>()));
#undef MLD_STRIP_ATTR8S
  (void)hipFuncSetAttribute((const void*)gemm_strip_kernel<0, 1, true, PREC_F32, 0, 1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (strip_lds_bytes<1, 1>()));
  (void)hipGetLastError();
#endif
  *out = e;
  return MLDHIP_OK;
}

void mldhip_destroy(mldhip_handle* e) {
  if (!e) return;
  DeviceGuard dg(e->device);
#if !defined(MLDHIP_SIM)
  (void)hipDeviceSynchronize();      // calls may still be in flight on other streams; their buffers are freed below
  for (auto& x : e->ctxs) {
    for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : x.step_graphs) (void)hipGraphExecDestroy(kv.second);
    if (x.done) (void)hipEventDestroy(x.done);
    if (x.loop_done) (void)hipEventDestroy(x.loop_done);
    if (x.pre_done) (void)hipEventDestroy(x.pre_done);
  }
  if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
  if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
  if (e->prep_stream) (void)hipStreamDestroy(e->prep_stream);
  if (e->many_start) (void)hipEventDestroy(e->many_start);
#endif
  if (e->arena) (void)hipFree(e->arena);
  if (e->arena_x3) (void)hipFree(e->arena_x3);
  if (e->ffn_streams) (void)hipFree(e->ffn_streams);
  if (e->loop_stream) (void)hipFree(e->loop_stream);
  if (e->loop_stream_x3) (void)hipFree(e->loop_stream_x3);
  if (e->cl_stream) (void)hipFree(e->cl_stream);
  if (e->cl_wave_off_dev) (void)hipFree(e->cl_wave_off_dev);
  if (e->loop_small) (void)hipFree(e->loop_small);
  for (auto& x : e->ctxs) {
    if (x.ws) (void)hipFree(x.ws);
    if (x.lens) (void)hipFree(x.lens);
    if (x.lens2) (void)hipFree(x.lens2);
    if (x.labels) (void)hipFree(x.labels);
  }
  if (e->trace_buf) (void)hipFree(e->trace_buf);
  if (e->nonfinite) (void)hipFree(e->nonfinite);
#if !defined(MLDHIP_SIM)
  if (e->cl_host_status) (void)hipHostFree(e->cl_host_status);
#else
  delete e->cl_host_status;
#endif
  delete e;
}

int mldhip_load_tensor(mldhip_handle* e, const char* key, const void* data, const int64_t* shape, int32_t ndim,
                       int32_t dtype, int32_t src_is_device) {
  if (!e || !key || !data || (!shape && ndim > 0)) return e ? e->fail(MLDHIP_EINVAL, "null argument") : MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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

int mldhip_set_option(mldhip_handle* e, const char* name, int64_t value) {
  if (!e || !name) return MLDHIP_EINVAL;
  const std::string n = name;
  if (n == "loop_kernel") {
    if (value < 0 || value > 4) return e->fail(MLDHIP_EINVAL, "loop_kernel must be 0 (auto), 1 (latency), 2 (throughput), 3 (sample-major persistent loop) or 4 (cluster loop)");
    if (value == 3 && e->finalized && !e->loop_ips) return e->fail(MLDHIP_EINVAL, "loop_kernel 3: the sample-major loop is built for fp32 loop arithmetic, ff_size 1024, 4 heads");
    if (value == 4 && e->finalized && !e->cl_stream) return e->fail(MLDHIP_EINVAL, "loop_kernel 4: the cluster loop is built for the split-f16 mode, latent_dim 256, ff_size 1024, 4 heads");
    e->loop_kernel = (int)value;
    if (value == 4) { e->cluster_failed = 0; if (e->cl_host_status) *e->cl_host_status = 0u; }
  } else if (n == "fused_x3") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "fused_x3 must be 0 or 1");
    e->fused_x3 = (int)value;
  } else if (n == "cluster_max_batch") {
    if (value < 0 || value > kClMaxCall) return e->fail(MLDHIP_EINVAL, "cluster_max_batch must be 0 .. %d", kClMaxCall);
    e->cluster_max_batch = (int)value;
  } else if (n == "cluster_wt") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "cluster_wt must be 0 (plain payload stores inside an XCD, write-through across) or 1 (write-through always)");
    e->cluster_wt = (int)value;
  } else if (n == "cluster_groups") {
    if (value != 0 && value != 4 && value != 8) return e->fail(MLDHIP_EINVAL, "cluster_groups must be 0 (auto: 8 up to 64 motions, 4 above), 4 or 8");
    e->cluster_groups = (int)value;
#if defined(MLDHIP_HOOKS)
  } else if (n == "cluster_inject") {
    // hooks build only: fault injection for the tests of the bounded waits -- member value - 1 of every cluster never raises its first flag and the wait bound shrinks to
    // 2 ms (GPU) / 1 500 polls (simulator); 0 = off
    if (value < 0 || value > kClMembersMax) return e->fail(MLDHIP_EINVAL, "cluster_inject must be 0 (off) or 1 + a member index");
    e->cluster_mute = (int)value - 1;
#if defined(MLDHIP_SIM)
    e->cluster_timeout = value ? 1500 : 0;
#else
    e->cluster_timeout = value ? 200000 : 0;
#endif
  } else if (n == "cluster_stale") {
    e->cluster_stale = value != 0;      // hooks build only: the next cluster launches find a stale epoch in a polled word (entry check)
  } else if (n == "cluster_chunk") {
    if (value < 8 || value > 8 * kClMaxClusters || value % 8) return e->fail(MLDHIP_EINVAL, "cluster_chunk must be a multiple of 8 in 8 .. %d", 8 * kClMaxClusters);
    e->cluster_chunk = (int)value;      // hooks build only: motions per cluster launch (tests of the several-launches path on a few motions)
  } else if (n == "cluster_lane") {
    e->cluster_lane = value != 0;       // hooks build only: 0 = cluster calls of different streams are NOT ordered behind each other (reproduces the co-residency starvation: tools/two_streams.py)
  } else if (n == "cluster_graph") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "cluster_graph must be 0 (eager issue) or 1 (graphs); 2 (the memset-node clear that reproduced the r05 replay fault) was removed in round 6: the kernel's entry check covers the case");
    e->cluster_graph = value != 0;      // hooks build only
  } else if (n == "fused_dbg") {
    if (value != 0 && value != 5) return e->fail(MLDHIP_EINVAL, "fused_dbg must be 0 or 5 (phase counters; the builds with wrong results live in tools/loopbench only)");
    if (value == 5 && !e->trace_buf && hipMalloc((void**)&e->trace_buf, (size_t)512 * 8 * 8 * sizeof(uint64_t)) != hipSuccess) return e->fail(MLDHIP_EHIP, "hipMalloc(trace)");
    e->fused_dbg = (int)value;
#endif
  } else if (n == "range_probe") {
    if (value < 0 || value > 2) return e->fail(MLDHIP_EINVAL, "range_probe must be 0 (off), 1 (seeded probe batch at finalize) or 2 (1 + the reverse-loop probe again on the caller's first batch)");
    e->range_probe = (int)value;
    e->finalized = false;            // the probe is part of finalize
  } else if (n == "fused_min_batch") {
    if (value < 0) return e->fail(MLDHIP_EINVAL, "fused_min_batch must be >= 0 (0 = automatic)");
    e->fused_min_batch = (int)std::min<int64_t>(value, 1 << 30);
  } else if (n == "strip_min_rows") {
    if (value < 1) return e->fail(MLDHIP_EINVAL, "strip_min_rows must be >= 1");
    e->strip_min_rows = (int)std::min<int64_t>(value, 1 << 30);
  } else if (n == "flash_attn") {
    if (value < 0 || value > 2) return e->fail(MLDHIP_EINVAL, "flash_attn must be 0 (never), 1 (auto) or 2 (always)");
    e->flash_attn = (int)value;
  } else if (n == "dec_tail") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "dec_tail must be 0 or 1");
    e->dec_tail = (int)value;
  } else if (n == "dec_l0_once") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "dec_l0_once must be 0 or 1");
    e->dec_l0_once = (int)value;
  } else if (n == "many_pipeline") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "many_pipeline must be 0 (one chain over all motions of a mldhip_sample_many call) or 1 (request after request, decodes on the side stream)");
    if (value && e->ctxs.size() < 2) return e->fail(MLDHIP_EINVAL, "many_pipeline needs two workspaces: create the handle with max_in_flight >= 2");
    e->many_pipeline = (int)value;
    return MLDHIP_OK;                   // (host-side orchestration only: nothing a captured graph bakes in changes -- the graphs stay)
  } else if (n == "dec_half") {
    if (value < 0 || value > 6 || value == 3 || value == 5) return e->fail(MLDHIP_EINVAL, "dec_half must be 0 (fp32 Q|K|V, split x3 products), 1 (half Q|K|V; strip height by launch size), 4 or 6 (1 with 64- / 96-row in-projection strips always) or 2 (1, but never overruled by finalize's probe)");
    // the probe's reading of the form is part of finalize: switching it on (with the veto in force) on a probed handle that has not read it asks for finalize again
    if (value != 0 && value != 2 && e->finalized && e->range_probe && e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->probe_err_decode >= 0.f && e->probe_err_decode_half < 0.f) e->finalized = false;
    e->dec_half = (int)value;
  } else if (n == "tile_x3") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "tile_x3 must be 0 or 1");
    e->tile_x3 = (int)value;
  } else if (n == "cross_fold") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "cross_fold must be 0 or 1");
    e->cross_fold = (int)value;
  } else if (n == "gemm_pipe") {
    if (value < 0 || value > 2) return e->fail(MLDHIP_EINVAL, "gemm_pipe must be 0 (off), 1 (auto: launches of >= 2 048 rows) or 2 (always)");
    e->gemm_pipe = (int)value;
  } else if (n == "strip_gemm") {
    if (value != 0 && value != 1) return e->fail(MLDHIP_EINVAL, "strip_gemm must be 0 or 1");
    e->strip_gemm = (int)value;
  } else if (n == "ffn_strip") {
    if (value != 0 && value != 1 && value != 3 && value != 4 && value != 6) return e->fail(MLDHIP_EINVAL, "ffn_strip must be 0 (off), 1 (auto: 64- or 96-row strips by launch size), 3, 4 or 6");
    e->ffn_strip = (int)value;
  } else if (n == "gemm_small_m") {
    if (value < 0) return e->fail(MLDHIP_EINVAL, "gemm_small_m must be >= 0");
    e->small_m = (int)std::min<int64_t>(value, 1 << 30);
  } else {
    return e->fail(MLDHIP_EINVAL, "unknown option %s", name);
  }
#if !defined(MLDHIP_SIM)
  // captured graphs bake the kernel choice in: drop them (nothing may be in flight on a context while it is rebuilt)
  DeviceGuard dg(e->device);
  for (auto& x : e->ctxs) {
    drain_context(x);
    for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
    x.graphs.clear();
    x.graph_lru.clear();
    for (auto& kv : x.step_graphs) (void)hipGraphExecDestroy(kv.second);
    x.step_graphs.clear();
  }
#endif
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

}  // extern "C"

namespace {
int denoiser_forward_impl(mldhip_handle* e, const float* sample_dev, int32_t timestep, const float* text_emb_dev,
                          const int32_t* actions_host, int32_t R, float* out_dev, void* stream_);

// Range probe of the F16X3 mode (include/mldhip.h "Range contract"): the split-f16 kernels against the exact-fp32 ones of the SAME
// handle on one seeded probe batch; a stage that disagrees (or is not finite) is switched to the fp32 kernels.
int range_probe(mldhip_handle* e, hipStream_t stream, const float* user_text = nullptr, const float* user_lat = nullptr, int user_B = 0) {
  const int D = e->cfg.latent_dim, NF = e->cfg.nfeats, TD = e->cfg.text_dim;
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto uni = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) + 1) * (1.0f / 16777217.0f); };
  auto fill = [&](std::vector<float>& v, float scale) {      // Box-Muller, seeded: the probe is a function of the weights only
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      const float r = std::sqrt(-2.0f * std::log(uni())), a = 6.283185307179586f * uni();
      v[i] = scale * r * std::cos(a); v[i + 1] = scale * r * std::sin(a);
    }
  };
  struct Dev {
    float* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    int up(const std::vector<float>& h) { return hipMalloc((void**)&p, h.size() * sizeof(float)) == hipSuccess && hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }
    int make(size_t n) { return hipMalloc((void**)&p, n * sizeof(float)) == hipSuccess && hipMemset(p, 0, n * sizeof(float)) == hipSuccess ? 0 : 1; }
  };
  auto rel_err = [](const std::vector<float>& a, const std::vector<float>& b) {       // max|a - b| / max|b|; inf when anything is not finite
    float d = 0.f, m = 0.f;
    for (size_t i = 0; i < a.size(); ++i) {
      if (!std::isfinite(a[i]) || !std::isfinite(b[i])) return std::numeric_limits<float>::infinity();
      d = std::max(d, std::fabs(a[i] - b[i])); m = std::max(m, std::fabs(b[i]));
    }
    return m > 0.f ? d / m : (d > 0.f ? std::numeric_limits<float>::infinity() : 0.f);
  };
  auto down = [&](const float* dev, size_t n, std::vector<float>& h) {
    h.resize(n);
    return hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
  };
  // user_text / user_lat: "range_probe" 2 -- the reverse-loop part once more on the first min(8, B) motions of the caller's first batch (device pointers of
  // a text-conditioned mldhip_sample call: [2B][TD] embeddings, unconditional half first, and [B][D] start latents); the decoder part is not repeated
  const bool user = user_text != nullptr && user_lat != nullptr && user_B > 0;
  const int Bp = user ? std::min(8, user_B) : std::min(8, e->cfg.max_batch);
  if (e->group_ready[0] && !is_novae(e)) {
    // ---- reverse loop.  (a) one denoiser call of the latency kernels at the first and the last timestep of the schedule
    std::vector<float> hs((size_t)2 * Bp * D), ht((size_t)2 * Bp * TD);
    fill(hs, 1.0f); fill(ht, 0.5f);
    if (user) {
      if (hipStreamSynchronize(stream) != hipSuccess ||
          hipMemcpy(hs.data(), user_lat, (size_t)Bp * D * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(ht.data(), user_text, (size_t)Bp * TD * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(ht.data() + (size_t)Bp * TD, user_text + (size_t)user_B * TD, (size_t)Bp * TD * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return e->fail(MLDHIP_EHIP, "range probe: copy of the caller's batch");
    }
    for (int i = 0; i < Bp * D; ++i) hs[(size_t)Bp * D + i] = hs[i];                      // both CFG halves see the same latents
    std::vector<int32_t> act((size_t)2 * Bp);
    for (int i = 0; i < 2 * Bp; ++i) act[i] = i % std::max(1, e->cfg.nclasses);
    Dev sample, text, out;
    if (sample.up(hs) || text.up(ht) || out.make((size_t)2 * Bp * D)) return e->fail(MLDHIP_EHIP, "range probe: hipMalloc");
    float worst = 0.f;
    std::vector<float> ha, hb;
    const int n = e->cfg.num_inference_steps;
    for (int which = 0; which < 2; ++which) {
      const int t = e->timesteps[which == 0 ? 0 : n - 1];
      for (int split = 1; split >= 0; --split) {
        e->split_loop_ok = split != 0;
        if (int rc = denoiser_forward_impl(e, sample.p, t, is_action(e) ? nullptr : text.p, is_action(e) ? act.data() : nullptr, 2 * Bp, out.p, stream)) return rc;
        if (down(out.p, (size_t)2 * Bp * D, split ? ha : hb)) return e->fail(MLDHIP_EHIP, "range probe: copy");
      }
      worst = std::max(worst, rel_err(ha, hb));
    }
    // (b) two reverse steps of the persistent loop (its operand images are not clamped: an overflow shows up as NaN here)
    // (run whenever the split stream exists: "fused_x3" / "tile_x3" / "loop_kernel" may be changed after finalize, and the verdict must cover them)
    if (e->loop_ips > 0 && e->loop_stream_x3) {
      const int fx3 = e->fused_x3;
      e->fused_x3 = 1;
      struct Restore { mldhip_handle* e; int v; ~Restore() { e->fused_x3 = v; } } restore{e, fx3};
      CtxUse use(e, stream);
      if (use.rc) return use.rc;
      Ctx c{e, stream};
      e->phase = 0;
      const float guidance = e->cfg.guidance_scale > 1.0f ? e->cfg.guidance_scale : 1.0f;
      for (int split = 1; split >= 0; --split) {
        e->split_loop_ok = split != 0;
        if (is_action(e)) {
          HIP_TRY(e, hipMemcpyAsync(e->labels_dev, act.data(), act.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
          action_rows(c, 2 * Bp, Bp, e->TP);
        } else {
          text_projection(c, text.p, 2 * Bp, e->TP);
        }
        launch_fused_loop(c, sample.p, Bp, std::min(2, n), guidance);
        if (c.rc) return c.rc;
        if (down(e->lat, (size_t)Bp * D, split ? ha : hb)) return e->fail(MLDHIP_EHIP, "range probe: copy");
      }
      // measured against the UPDATE the two steps made (latents - start noise), not against the latents: near t = T a DDIM step moves
      // x by a few per cent, and how much depends on the schedule; the update is (guided eps) x (step coefficients), so this reads
      // like (a) times the guidance amplification (2 g - 1 at worst) -- hence the factor on the tolerance
      float d = 0.f, m = 0.f;
      bool finite = true;
      for (size_t i = 0; i < ha.size(); ++i) {
        finite = finite && std::isfinite(ha[i]) && std::isfinite(hb[i]);
        d = std::max(d, std::fabs(ha[i] - hb[i]));
        m = std::max(m, std::fabs(hb[i] - hs[i]));
      }
      const float amp = std::max(1.0f, 2.0f * guidance - 1.0f);
      const float eb = !finite ? std::numeric_limits<float>::infinity() : (m > 0.f ? d / m / amp : (d > 0.f ? std::numeric_limits<float>::infinity() : 0.f));
      worst = std::max(worst, eb);
      // (c) the cluster loop (kernels/loop_cluster.hpp: split-f16 only, unclamped images like the persistent loop's) on the same two steps, against the exact-fp32 result `hb`
#if defined(MLDHIP_SIM)
      const bool probe_cluster = e->cl_stream && (e->loop_kernel == 4 || e->cluster_max_batch > 0);      // (the simulator's handles never pick it by themselves: 20 s per probe saved)
#else
      const bool probe_cluster = e->cl_stream != nullptr;
#endif
      if (probe_cluster) {
        e->split_loop_ok = true;
        // both forms (advisor r5): 8 column groups per token (24 workgroups per cluster: calls of up to 64 motions -- what a probe batch of 8 picks by itself) and 4 (12 workgroups:
        // calls of 65 .. 256 motions); they differ in how linear1 / linear2 / the skip linear are split over members and waves, i.e. in the order of sums
        const int cg_saved = e->cluster_groups;
        struct RestoreCG { mldhip_handle* e; int v; ~RestoreCG() { e->cluster_groups = v; } } restore_cg{e, cg_saved};
        const int first = cluster_groups(e, Bp);      // what the handle picks for the probe batch: 8 unless the device is small or the option says 4
        for (int form = 0; form < 2 && !e->cluster_failed; ++form) {
          if (form == 1) {
            e->cluster_groups = first == 8 ? 4 : 8;
            if (cluster_groups(e, Bp) == first) break;      // the other form is not available on this device: nothing new to run
          }
          std::vector<float> hc_;
          {
            ClusterLane lane(e, c.stream, true);
            launch_cluster_loop(c, sample.p, Bp, std::min(2, n), guidance);
          }
          if (c.rc) return c.rc;
          if (down(e->lat, (size_t)Bp * D, hc_)) return e->fail(MLDHIP_EHIP, "range probe: copy");
          if (cluster_timed_out(e)) {
            // the device did not keep the launch's workgroups resident together (a wait ran into its 200 ms bound): not an arithmetic verdict -- the handle leaves the cluster loop
            e->cluster_failed = 1;
            if (e->cl_host_status) *e->cl_host_status = 0u;
          } else {
            float d2 = 0.f;
            bool fin2 = true;
            for (size_t i = 0; i < hc_.size(); ++i) { fin2 = fin2 && std::isfinite(hc_[i]); d2 = std::max(d2, std::fabs(hc_[i] - hb[i])); }
            worst = std::max(worst, !fin2 ? std::numeric_limits<float>::infinity() : (m > 0.f ? d2 / m / amp : (d2 > 0.f ? std::numeric_limits<float>::infinity() : 0.f)));
          }
        }
      }
    }
    if (user) worst = std::max(worst, e->probe_err_loop);          // the verdict covers the seeded batch AND the caller's
    e->probe_err_loop = worst;
    e->split_loop_ok = worst <= MLDHIP_PROBE_TOL;
  }
  if (user) { e->phase = 0; return MLDHIP_OK; }
  // The probe must run the kernels production calls run.  The row-strip GEMMs, the fused decoder tail, the final strip and (diffusion-only
  // variant) the pipelined 128 x 256 tile are selected by row count ("gemm_small_m", gemm_pipe_min_rows) and the two attention forms by
  // the number of (sample, head) pairs -- a probe batch is far below all of these (advisor r4: at 4 x 64 = 256 rows both arms of the decoder
  // probe ran the SAME fp32 small-M kernel for every GEMM but two).  For the duration of the probe the thresholds are lifted, and the decode
  // is probed once per attention form; what stays unprobed is listed in include/mldhip.h "Range contract".
  struct Lift {
    mldhip_handle* e; int small_m, pipe_rows, flash;
    explicit Lift(mldhip_handle* e_) : e(e_), small_m(e_->small_m), pipe_rows(e_->gemm_pipe_min_rows), flash(e_->flash_attn) { e->small_m = 0; e->gemm_pipe_min_rows = 0; }
    ~Lift() { e->small_m = small_m; e->gemm_pipe_min_rows = pipe_rows; e->flash_attn = flash; }
  };
  if (e->group_ready[1] && !is_novae(e)) {
    // ---- decoder: decodes of 4 motions x min(64, max_frames) frames (two full, two ragged), key-blocked and whole-K/V attention
    Lift lift(e);
    const int B = std::min(4, e->cfg.max_batch), T = std::min(64, e->cfg.max_frames);
    std::vector<float> hz((size_t)B * D);
    fill(hz, 4.0f);
    std::vector<int32_t> lens(B, T);
    if (B > 1) lens[1] = std::max(1, T - 7);
    if (B > 3) lens[3] = std::max(1, T / 2 + 1);
    Dev z, feats;
    if (z.up(hz) || feats.make((size_t)B * T * NF)) return e->fail(MLDHIP_EHIP, "range probe: hipMalloc");
    std::vector<float> ha, hb;
    auto run = [&](std::vector<float>& h) -> int {
      CtxUse use(e, stream);
      if (use.rc) return use.rc;
      HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lens.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      Ctx c{e, stream};
      e->phase = 1;
      decode_body(c, z.p, B, T, feats.p);
      if (c.rc) return c.rc;
      if (down(feats.p, (size_t)B * T * NF, h)) return e->fail(MLDHIP_EHIP, "range probe: copy");
      return 0;
    };
    const int dh = e->dec_half;
    struct RestoreDH { mldhip_handle* e; int v; ~RestoreDH() { e->dec_half = v; } } restore_dh{e, dh};
    e->split_decode_ok = false;                    // the exact-fp32 decode: the reference of every form below
    if (int rc = run(hb)) return rc;
    e->split_decode_ok = true;
    float worst = 0.f;
    for (int form = 0; form < 2; ++form) {         // fp32 Q | K | V, split x3 products: key-blocked and whole-K/V attention
      e->dec_half = 0;
      e->flash_attn = form == 0 ? 2 : 0;
      if (int rc = run(ha)) return rc;
      worst = std::max(worst, rel_err(ha, hb));
    }
    float half_err = -1.f;
    if (dh) {
      // the opt-in self-attention block on half Q | K | V (kernels/dec_half.hpp): its own bound, read on UNIT-normal latents -- with large latents the per-sample
      // cross-attention vector drowns the frame-to-frame signal the self-attention carries and the form looks 10-30x better than it is (profiles/r06_decoder_precision.json)
      std::vector<float> hz1((size_t)B * D);
      fill(hz1, 1.0f);
      HIP_TRY(e, hipMemcpy(z.p, hz1.data(), hz1.size() * sizeof(float), hipMemcpyHostToDevice));
      e->dec_half = 0;
      e->split_decode_ok = false;
      if (int rc = run(hb)) return rc;
      e->split_decode_ok = true;
      e->dec_half = 2;
      if (int rc = run(ha)) return rc;
      half_err = rel_err(ha, hb);
    }
    e->probe_err_decode = worst;
    e->split_decode_ok = worst <= MLDHIP_PROBE_TOL;
    e->probe_err_decode_half = half_err;
    e->dec_half_ok = !dh || (half_err >= 0.f && half_err <= MLDHIP_PROBE_TOL_HALF);      // (option off: nothing to veto; switching it on later un-finalizes the handle, mldhip_set_option)
  }
  if (e->group_ready[0] && is_novae(e)) {
    // ---- diffusion-only variant: one denoiser call (every GEMM and the frame-level attention run split in this mode) on 4 CFG rows x 128
    //      frames, on the pipelined tile + key-blocked head-dim-128 attention and on the staged tile + two-phase attention
    Lift lift(e);
    const int R = 2 * std::min(2, e->cfg.max_batch), T = std::min(128, e->cfg.max_frames);
    std::vector<float> hx((size_t)R * T * NF), ht((size_t)R * TD);
    fill(hx, 1.0f); fill(ht, 0.5f);
    std::vector<int32_t> lens(R, T);
    lens[1] = std::max(1, T - 5); lens[R - 1] = std::max(1, T - 5);
    Dev x, text, out;
    if (x.up(hx) || text.up(ht) || out.make((size_t)R * T * NF)) return e->fail(MLDHIP_EHIP, "range probe: hipMalloc");
    std::vector<float> ha, hb;
    float worst = 0.f;
    const int pipe = e->gemm_pipe;
    for (int form = 0; form < 2; ++form) {
      e->flash_attn = form == 0 ? 2 : 0;
      e->gemm_pipe = form == 0 ? pipe : 0;
      int rc = 0;
      for (int split = 1; split >= 0 && !rc; --split) {
        e->split_decode_ok = split != 0;
        rc = mldhip_denoiser_forward_novae(e, x.p, e->timesteps[0], text.p, lens.data(), R, T, out.p, stream);
        if (!rc && down(out.p, (size_t)R * T * NF, split ? ha : hb)) rc = e->fail(MLDHIP_EHIP, "range probe: copy");
      }
      e->gemm_pipe = pipe;
      if (rc) return rc;
      worst = std::max(worst, rel_err(ha, hb));
    }
    e->probe_err_decode = worst;
    e->split_decode_ok = worst <= MLDHIP_PROBE_TOL;
  }
  e->phase = 0;
  return MLDHIP_OK;
}
}  // namespace

extern "C" {
int mldhip_finalize_weights(mldhip_handle* e, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  if (e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE) {
    // the staged GEMMs of these modes run on split-bf16 MFMAs: split the weights once, here, not in every workgroup
    if (!e->arena_x3 && hipMalloc((void**)&e->arena_x3, e->arena_floats * sizeof(float)) != hipSuccess) return e->fail(MLDHIP_EHIP, "hipMalloc(split weights)");
    const long long groups = (long long)((e->arena_floats + 31) / 32);   // arena_floats is a multiple of kAlign = 64
    MLD_LAUNCH(split_bf16_weights_kernel, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, stream, e->arena, e->arena_x3, groups);
    if (check_launch(c, "split_bf16_weights")) return c.rc;
  }
  if (int rc = build_loop_stream(c)) return rc;
  if (int rc = build_cluster_stream(c)) return rc;
  if (int rc = build_ffn_streams(c)) return rc;
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
      novae_fold_memory(c, e->TKV, n, (long long)n * 2 * D, e->TKW, e->TKU, e->TKC);      // ... and so do their folded forms ("cross_fold")
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
    x.graph_lru.clear();
    for (auto& kv : x.step_graphs) (void)hipGraphExecDestroy(kv.second);
    x.step_graphs.clear();
    x.used = false;
  }
#endif
  bind_context(e, 0);
  e->next_ctx = 0;
  if (e->loop_kernel == 4 && !e->cl_stream)
    return e->fail(MLDHIP_EINVAL, "loop_kernel 4: the cluster loop is built for the split-f16 mode, latent_dim 256, ff_size 1024, 4 heads");
  if (e->loop_kernel == 3 && !e->loop_ips)      // (set before finalize: refused here, like mldhip_set_option refuses it afterwards)
    return e->fail(MLDHIP_EINVAL, "loop_kernel 3: the sample-major loop is built for fp32 / split-f16 loop arithmetic, latent_dim 256, ff_size 1024, 4 heads");
  e->finalized = true;
  e->split_loop_ok = e->split_decode_ok = e->dec_half_ok = true;
  e->probe_err_loop = e->probe_err_decode = e->probe_err_decode_half = -1.f;
  if (e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->range_probe) {
    if (int rc = range_probe(e, stream)) { e->finalized = false; return rc; }
    e->probe_first_call = e->range_probe == 2;
  }
  return MLDHIP_OK;
}

int mldhip_numeric_status(mldhip_handle* e, mldhip_numeric_info* out) {
  if (!e || !out) return MLDHIP_EINVAL;
  if (out->struct_size != (int32_t)sizeof(mldhip_numeric_info)) return e->fail(MLDHIP_EINVAL, "mldhip_numeric_info.struct_size mismatch (ABI)");
  DeviceGuard dg(e->device);
  HIP_TRY(e, hipDeviceSynchronize());
  unsigned n = 0;
  HIP_TRY(e, hipMemcpy(&n, e->nonfinite, sizeof n, hipMemcpyDeviceToHost));
  HIP_TRY(e, hipMemset(e->nonfinite, 0, sizeof n));
  // a cluster launch that ran into its wait bound poisoned its latents (counted above) and left its status word set: the handle stays off the cluster loop from here on
  // (the captured graphs that hold it are dropped); mldhip_set_option("loop_kernel", 4) re-arms it
  if (!e->cluster_failed && cluster_timed_out(e)) {
    e->cluster_failed = 1;
    if (e->cl_host_status) *e->cl_host_status = 0u;
#if !defined(MLDHIP_SIM)
    for (auto& x : e->ctxs) {
      drain_context(x);                          // (torch's streams are non-blocking: the device-wide sync above is what orders this, the drain says so explicitly -- advisor r5)
      for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
      x.graphs.clear();
      x.graph_lru.clear();
    }
#endif
  }
  out->probed = e->probe_err_loop >= 0.f || e->probe_err_decode >= 0.f;
  out->loop_split_ok = e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->split_loop_ok;
  out->decode_split_ok = e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->split_decode_ok;
  out->probe_err_loop = e->probe_err_loop;
  out->probe_err_decode = e->probe_err_decode;
  out->nonfinite_values = (int64_t)n;
  out->cluster_loop = !e->cl_stream ? 0 : e->cluster_foreign ? 3 : e->cluster_failed ? 2 : 1;
  out->reserved = 0;
  out->decode_half_ok = e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->split_decode_ok && e->dec_half && (e->dec_half_ok || e->dec_half == 2);
  out->probe_err_decode_half = e->probe_err_decode_half;
  return MLDHIP_OK;
}

}  // extern "C"

namespace {
#if !defined(MLDHIP_SIM)
// The captured graph of (B, Tmax, requested outputs) on the bound workspace context: looked up, or captured now.  Graphs
// read the engine's staging buffers (text_in / lat_in / labels / lens) and write lat / feats_int / joints_int, so they
// are independent of the caller's buffers and of how many requests make up the B motions.
int graph_for(mldhip_handle* e, const GraphKey& key, bool text_condition, hipGraphExec_t* out) {
  auto& graphs = e->ctxs[e->cur_ctx].graphs;
  auto& lru = e->ctxs[e->cur_ctx].graph_lru;
  auto same = [&](const GraphKey& k) { return !(k < key) && !(key < k); };
  lru.erase(std::remove_if(lru.begin(), lru.end(), same), lru.end());
  lru.push_back(key);
  auto it = graphs.find(key);
  if (it == graphs.end()) {
    // one graph per (B, Tmax, outputs): a serving loop with ragged batches sees many Tmax values, so keep a generous
    // number (each exec holds ~2 100 kernel nodes, a few MB) and evict the least recently used one beyond it
    while (graphs.size() >= kGraphCacheCapacity) {
      auto victim = graphs.find(lru.front());
      lru.erase(lru.begin());
      if (victim == graphs.end()) continue;
      drain_context(e->ctxs[e->cur_ctx]);               // the victim may still be replaying on this context's last stream
      (void)hipGraphExecDestroy(victim->second);
      graphs.erase(victim);
    }
    hipGraph_t graph = nullptr;
    HIP_TRY(e, hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
    int rc = 0;
    if (key.dec_only) {
      Ctx cd{e, e->cap_stream};
      enqueue_decode(cd, key.B, key.T, key.feats ? e->feats_int : nullptr, key.joints ? e->joints_int : nullptr);
      rc = cd.rc;
    } else {
      e->sample_part = key.part;
      rc = enqueue_sample(e, e->cap_stream, text_condition ? e->text_in : nullptr, e->lat_in, key.B, key.T, nullptr,
                          key.feats ? e->feats_int : nullptr, key.joints ? e->joints_int : nullptr);
      e->sample_part = 0;
    }
    hipError_t s = hipStreamEndCapture(e->cap_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(s));
    hipGraphExec_t exec = nullptr;
    s = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(s));
    it = graphs.emplace(key, exec).first;
  }
  *out = it->second;
  return MLDHIP_OK;
}
#endif

// Self-healing of the cluster loop (advisor r5): the kernel sets a pinned host word next to its sticky status word when a wait runs into its bound.  Every sample call looks
// at it first -- a plain host read, no device synchronisation: a handle whose cluster launch timed out (its latents were poisoned with NaN and counted) serves the NEXT call
// on the other loop families already, without waiting for the caller to poll mldhip_numeric_status.  The graphs that hold the kernel are dropped (failure path: blocking).
void heal_cluster(mldhip_handle* e) {
  if (!e->cl_host_status || e->cluster_failed || *reinterpret_cast<volatile unsigned*>(e->cl_host_status) == 0u) return;
  e->cluster_failed = 1;
  *e->cl_host_status = 0u;
#if !defined(MLDHIP_SIM)
  for (auto& x : e->ctxs) {
    drain_context(x);
    for (auto& kv : x.graphs) (void)hipGraphExecDestroy(kv.second);
    x.graphs.clear();
    x.graph_lru.clear();
  }
#endif
  (void)cluster_timed_out(e);          // the device-side sticky words have been acted on: cleared (a later mldhip_numeric_status must not fail a re-armed handle for them)
}

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
  heal_cluster(e);
  if (e->probe_first_call && text_emb_dev) {              // "range_probe" 2: the loop probe on THIS batch before it is sampled (one-off, synchronous)
    e->probe_first_call = false;
    const bool was_ok = e->split_loop_ok;
    if (was_ok) {
      if (int rc = range_probe(e, stream, text_emb_dev, init_latents_dev, B)) return rc;
    }
  }
  CtxUse use(e, stream);                                  // picks + binds a workspace context (see WsContext)
  if (use.rc) return use.rc;
  ClusterLane lane(e, stream, e->cluster_lane && use_cluster(e, B));         // cluster launches never side by side (engine/params.hpp)
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
  // Calls served by the cluster loop replay a captured graph like the rest; its flags are cleared by a kernel, not a memset node: replays of
  // a hipMemsetAsync node in front of den_cluster_kernel left address-like words in the tail of the buffer on this runtime (r05).
  if (e->cfg.use_graph && (!use_cluster(e, B) || e->cluster_graph)) {    // cluster_graph: true outside the hooks build
    const size_t D = e->cfg.latent_dim, NF = e->cfg.nfeats;
    const bool want_j = joints_out_dev != nullptr, want_f = feats_out_dev != nullptr || want_j;
    if (text_emb_dev)
      HIP_TRY(e, hipMemcpyAsync(e->text_in, text_emb_dev, (size_t)2 * B * e->cfg.text_dim * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(e, hipMemcpyAsync(e->lat_in, init_latents_dev, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    hipGraphExec_t exec = nullptr;
    if (int rc = graph_for(e, GraphKey{B, T, want_f, want_j}, text_emb_dev != nullptr, &exec)) return rc;
    HIP_TRY(e, hipGraphLaunch(exec, stream));
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

namespace {
// "many_pipeline": the requests of a mldhip_sample_many call ONE AFTER THE OTHER, each on the single-request path (the reverse loop of a request is one cluster
// launch, kernels/loop_cluster.hpp) -- the reference's own shape, batch after batch (mld.py:618-672, test.py:116-119) -- with the two halves of consecutive requests
// overlapped: the cluster launch holds 192 of 256 CUs at a few per cent of the matrix pipe for ~6.8 ms, the 44 decode launches of the previous request (one round
// of workgroups each on an idle chip) run beside it on the CUs it leaves free.
//   caller's stream S:  [wait ws(k) free] inputs(k) -> loop(k) -> record loop_done(k)                 ... after the last request: wait for every decode
//   side stream D:                                               wait loop_done(k) -> decode(k) -> outputs(k) -> record ws(k).done
// (requests of one cluster launch each -- every bs-64 request -- move inputs(k) and what precedes the launch to a third stream: the schedule in the body below.)
// Two workspaces alternate (request k + 2 waits for decode k).  D has the lowest stream priority: a cluster launch needs its workgroups resident together, the
// decode's workgroups are short and independent of it -- they can only delay it, and they end.  The lane (ClusterLane) is held for the whole call; its event is
// recorded on S behind the join.  Every request gets exactly what mldhip_sample gives it (same kernels, same graphs' machine code): bit-identical, tested.
int sample_many_pipelined(mldhip_handle* e, const mldhip_request* rq, int nreq, const std::vector<int32_t>& tmax, hipStream_t stream) {
  const bool action = is_action(e);
  const size_t D = e->cfg.latent_dim, NF = e->cfg.nfeats, TD = e->cfg.text_dim, NJ = (size_t)e->cfg.njoints * 3;
  ClusterLane lane(e, stream, e->cluster_lane);
#if !defined(MLDHIP_SIM)
  if (!e->side_stream) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIP_TRY(e, hipStreamCreateWithPriority(&e->side_stream, hipStreamNonBlocking, least));
  }
  hipStream_t side = e->side_stream;
#else
  hipStream_t side = stream;
#endif
  bool replay = false, split = false;
#if !defined(MLDHIP_SIM)
  replay = e->cfg.use_graph && e->cluster_graph;
  // Replayed calls whose requests are ONE cluster launch each get the tighter schedule (three streams):
  //   prep stream P:      [ws(k+1): loop k-1 and decode k-1 done] inputs(k+1) -> condition rows + flag clear (a graph, GraphKey.part 1) -> record pre_done(k+1)
  //   caller's stream S:  wait pre_done(k) -> cluster launch(k) (issued directly) -> record loop_done(k)
  //   side stream D:      wait loop_done(k) -> non-finite count, latents out, lengths -> decode(k) (a graph) -> outputs(k) -> record ws(k).done
  // so on S the cluster kernels follow each other with two event packets between them, and the decode of request k (lowest priority) becomes ready at the same instant as
  // the launch of request k + 1.  (First form of the round: everything but the decode on S -- 76 us between consecutive cluster kernels, in which the decode's first dozen
  // kernels took the chip before the launch did: profiles/r06_trace_pipeline.log.)
  split = replay;
  for (int i = 0; i < nreq; ++i) split = split && use_cluster(e, rq[i].B) && rq[i].B <= e->cluster_chunk;
  if (split && !e->prep_stream) {
    HIP_TRY(e, hipStreamCreateWithFlags(&e->prep_stream, hipStreamNonBlocking));
    HIP_TRY(e, hipEventCreateWithFlags(&e->many_start, hipEventDisableTiming));
  }
#endif
  std::vector<int> used, ctx_of(nreq, 0);
  int rc = MLDHIP_OK;
#if !defined(MLDHIP_SIM)
  hipStream_t prep = split ? e->prep_stream : stream;
  if (split) {
    HIP_TRY(e, hipEventRecord(e->many_start, stream));
    HIP_TRY(e, hipStreamWaitEvent(prep, e->many_start, 0));          // the requests' inputs are the caller's stream's products
  }
  if (replay) for (int i = 0; i < nreq; ++i) {                        // two contexts alternate
    ctx_of[i] = int(e->next_ctx++ % e->ctxs.size());
    if (std::find(used.begin(), used.end(), ctx_of[i]) == used.end()) used.push_back(ctx_of[i]);
  }
  auto stage_pre = [&](int i) -> int {           // inputs of request i into its context + what precedes its cluster launch
    const mldhip_request& r = rq[i];
    WsContext& x = e->ctxs[ctx_of[i]];
    bind_context(e, ctx_of[i]);
    if (x.used) HIP_TRY(e, hipStreamWaitEvent(prep, x.done, 0));      // the context's last decode (request i - 2, or an earlier call)
    if (split && i >= 2) HIP_TRY(e, hipStreamWaitEvent(prep, x.loop_done, 0));   // ... and its last cluster launch (on S; nothing to wait for when prep IS S)
    const float* text = action ? nullptr : r.text_emb_dev;
    if (action) {
      HIP_TRY(e, hipMemsetAsync(e->labels_dev, 0, (size_t)r.B * sizeof(int32_t), prep));
      HIP_TRY(e, hipMemcpyAsync(e->labels_dev + r.B, r.actions_host, (size_t)r.B * sizeof(int32_t), hipMemcpyHostToDevice, prep));
    }
    if (text) HIP_TRY(e, hipMemcpyAsync(e->text_in, text, (size_t)2 * r.B * TD * sizeof(float), hipMemcpyDeviceToDevice, prep));
    HIP_TRY(e, hipMemcpyAsync(e->lat_in, r.init_latents_dev, (size_t)r.B * D * sizeof(float), hipMemcpyDeviceToDevice, prep));
    if (split) {
      hipGraphExec_t pre = nullptr;
      GraphKey kp{r.B, 0, false, false}; kp.part = 1;
      if (int rc2 = graph_for(e, kp, text != nullptr, &pre)) return rc2;
      HIP_TRY(e, hipGraphLaunch(pre, prep));
      HIP_TRY(e, hipEventRecord(x.pre_done, prep));
    }
    return MLDHIP_OK;
  };
  if (replay && (rc = stage_pre(0))) return rc;
#endif
  for (int i = 0; i < nreq && !rc; ++i) {
    const mldhip_request& r = rq[i];
    const int B = r.B, T = tmax[i];
    const bool want_j = r.joints_out_dev != nullptr, want_f = r.feats_out_dev != nullptr || want_j;
    const float* text = action ? nullptr : r.text_emb_dev;
    if (replay) {
#if !defined(MLDHIP_SIM)
      WsContext& x = e->ctxs[ctx_of[i]];
      bind_context(e, ctx_of[i]);
      hipGraphExec_t loop = nullptr, dec = nullptr;
      GraphKey kd{B, T, want_f, want_j}; kd.dec_only = true;
      if ((rc = graph_for(e, kd, text != nullptr, &dec))) break;
      if (split) {
        HIP_TRY(e, hipStreamWaitEvent(stream, x.pre_done, 0));
        // the launch itself is issued directly, not as a graph of one kernel (a graph launch puts its own packets in front of its first node)
        e->sample_part = 2;
        rc = enqueue_sample(e, stream, text ? e->text_in : nullptr, e->lat_in, B, 0, nullptr, nullptr, nullptr);
        e->sample_part = 0;
        if (rc) break;
      } else {
        GraphKey kl{B, T, false, false};
        if ((rc = graph_for(e, kl, text != nullptr, &loop))) break;
        HIP_TRY(e, hipGraphLaunch(loop, stream));
        if (r.latents_out_dev) HIP_TRY(e, hipMemcpyAsync(r.latents_out_dev, e->lat, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
      }
      HIP_TRY(e, hipEventRecord(x.loop_done, stream));
      if (i + 1 < nreq && (rc = stage_pre(i + 1))) break;                // request i + 1: on the prep stream beside this launch (split), or behind it on S
      bind_context(e, ctx_of[i]);
      HIP_TRY(e, hipStreamWaitEvent(side, x.loop_done, 0));
      if (split) {
        Ctx cs{e, side};
        count_nonfinite(cs, e->lat, (long long)B * D);
        if ((rc = cs.rc)) break;
        if (r.latents_out_dev) HIP_TRY(e, hipMemcpyAsync(r.latents_out_dev, e->lat, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, side));
      }
      // the lengths are the decode's alone (the latent loop has no masks): copied on the side stream, in order behind the decode that used this context last
      HIP_TRY(e, hipMemcpyAsync(e->lens_dev, r.lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, side));
      HIP_TRY(e, hipGraphLaunch(dec, side));
      if (r.feats_out_dev) HIP_TRY(e, hipMemcpyAsync(r.feats_out_dev, e->feats_int, (size_t)B * T * NF * sizeof(float), hipMemcpyDeviceToDevice, side));
      if (r.joints_out_dev) HIP_TRY(e, hipMemcpyAsync(r.joints_out_dev, e->joints_int, (size_t)B * T * NJ * sizeof(float), hipMemcpyDeviceToDevice, side));
      HIP_TRY(e, hipEventRecord(x.done, side));
      x.used = true;
#endif
    } else {
      // eager issue (no graphs: the simulator; hooks builds with "cluster_graph" 0): the two halves one behind the other per request, outputs straight into the caller's buffers
      const int k = int(e->next_ctx++ % e->ctxs.size());
      WsContext& x = e->ctxs[k];
#if !defined(MLDHIP_SIM)
      if (x.used) HIP_TRY(e, hipStreamWaitEvent(stream, x.done, 0));
#endif
      bind_context(e, k);
      if (std::find(used.begin(), used.end(), k) == used.end()) used.push_back(k);
      HIP_TRY(e, hipMemcpyAsync(e->lens_dev, r.lengths_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      if (action) {
        HIP_TRY(e, hipMemsetAsync(e->labels_dev, 0, (size_t)B * sizeof(int32_t), stream));
        HIP_TRY(e, hipMemcpyAsync(e->labels_dev + B, r.actions_host, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, stream));
      }
      if ((rc = enqueue_sample(e, stream, text, r.init_latents_dev, B, T, r.latents_out_dev, nullptr, nullptr))) break;
#if !defined(MLDHIP_SIM)
      HIP_TRY(e, hipEventRecord(x.loop_done, stream));
      HIP_TRY(e, hipStreamWaitEvent(side, x.loop_done, 0));
#endif
      Ctx cd{e, side};
      enqueue_decode(cd, B, T, r.feats_out_dev, r.joints_out_dev);
      rc = cd.rc;
#if !defined(MLDHIP_SIM)
      HIP_TRY(e, hipEventRecord(x.done, side));
      x.used = true;
#endif
    }
  }
#if !defined(MLDHIP_SIM)
  for (int k : used)
    if (e->ctxs[k].used) (void)hipStreamWaitEvent(stream, e->ctxs[k].done, 0);     // the caller's stream is ordered behind every decode of the call
#endif
  return rc;
}

// Several independent requests as ONE reverse-diffusion chain + ONE decode (mldhip_sample_many): inputs are gathered into
// the engine's staging buffers (unconditional halves first, as one big CFG batch), outputs scattered per request with
// each request's own Tmax as its row pitch.  Motions never interact (attention is per sample), so results equal the
// per-request calls up to the summation order of the kernel family picked for the larger row count.
int sample_many_impl(mldhip_handle* e, const mldhip_request* rq, int nreq, hipStream_t stream) {
  if (!e->finalized) return e->fail(MLDHIP_ESTATE, "mldhip_sample_many before mldhip_finalize_weights");
  heal_cluster(e);
  const bool action = is_action(e);
  bool want_j = false, want_f = false;
  int Btot = 0, T = 0;
  std::vector<int32_t> lens, tmax(nreq, 0);
  for (int i = 0; i < nreq; ++i) {
    const mldhip_request& r = rq[i];
    if (!r.init_latents_dev || (action ? !r.actions_host : !r.text_emb_dev)) return e->fail(MLDHIP_EINVAL, "request %d: null input pointer", i);
    if (r.joints_out_dev && is_actor(e)) return e->fail(MLDHIP_ESTATE, "joints of the ActorVae feature layout need SMPL (out of scope)");
    if (int rc = validate_lengths(e, r.lengths_host, r.B, &tmax[i])) return rc;
    if (action)
      for (int k = 0; k < r.B; ++k)
        if (r.actions_host[k] < 0 || r.actions_host[k] >= e->cfg.nclasses)
          return e->fail(MLDHIP_EINVAL, "request %d: actions[%d]=%d outside [0, nclasses=%d)", i, k, r.actions_host[k], e->cfg.nclasses);
    lens.insert(lens.end(), r.lengths_host, r.lengths_host + r.B);
    Btot += r.B;
    T = std::max(T, tmax[i]);
    want_j = want_j || r.joints_out_dev;
    want_f = want_f || r.feats_out_dev || r.joints_out_dev;
  }
  if (!e->group_ready[0] || !e->group_ready[1] || (want_j && !e->group_ready[2]))
    return e->fail(MLDHIP_ESTATE, "mldhip_sample_many needs denoiser.*, vae.decoder.* (and mean/std for joints) loaded");
  if (e->many_pipeline && nreq >= 2 && e->ctxs.size() >= 2) {
    bool ok = true;
    for (int i = 0; i < nreq; ++i) ok = ok && use_cluster(e, rq[i].B) && (rq[i].feats_out_dev || rq[i].joints_out_dev);
    if (ok) return sample_many_pipelined(e, rq, nreq, tmax, stream);
  }
  if (Btot > e->cfg.max_batch) return e->fail(MLDHIP_EINVAL, "requests hold %d motions, max_batch is %d", Btot, e->cfg.max_batch);
  CtxUse use(e, stream);
  if (use.rc) return use.rc;
  ClusterLane lane(e, stream, e->cluster_lane && use_cluster(e, Btot));
  const size_t D = e->cfg.latent_dim, NF = e->cfg.nfeats, TD = e->cfg.text_dim, NJ = (size_t)e->cfg.njoints * 3;
  HIP_TRY(e, hipMemcpyAsync(e->lens_dev, lens.data(), (size_t)Btot * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  if (action) {
    std::vector<int32_t> lab(2 * (size_t)Btot, 0);        // cond = cat(zeros_like(actions), actions) (mld.py:722-725)
    int o = 0;
    for (int i = 0; i < nreq; ++i) { std::copy(rq[i].actions_host, rq[i].actions_host + rq[i].B, lab.begin() + Btot + o); o += rq[i].B; }
    HIP_TRY(e, hipMemcpyAsync(e->labels_dev, lab.data(), lab.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  }
  Ctx cio{e, stream};
  {
    GatherArgs ga;
    int o = 0, bmax = 0;
    for (int i = 0; i < kMaxRequests; ++i) { ga.text[i] = nullptr; ga.lat[i] = nullptr; ga.off[i] = 0; ga.nb[i] = 0; }
    for (int i = 0; i < nreq; ++i) {
      ga.text[i] = action ? nullptr : rq[i].text_emb_dev; ga.lat[i] = rq[i].init_latents_dev; ga.off[i] = o; ga.nb[i] = rq[i].B;
      o += rq[i].B; bmax = std::max(bmax, (int)rq[i].B);
    }
    ga.text_in = e->text_in; ga.lat_in = e->lat_in; ga.Btot = Btot; ga.TD = (int)TD; ga.D = (int)D;
    const long long per = (action ? 0 : 2LL * bmax * (long long)TD) + (long long)bmax * (long long)D;      // elements of the largest request
    const unsigned chunks = (unsigned)std::min<long long>(64, std::max<long long>(1, (per + 2047) / 2048));
    MLD_LAUNCH(gather_requests_kernel, dim3((unsigned)nreq, chunks), dim3(256), 0, stream, ga);
    if (check_launch(cio, "gather_requests")) return cio.rc;
  }
  const float* text = action ? nullptr : e->text_in;
  bool replayed = false;
#if !defined(MLDHIP_SIM)
  if (e->cfg.use_graph && (!use_cluster(e, Btot) || e->cluster_graph)) {
    hipGraphExec_t exec = nullptr;
    if (int rc = graph_for(e, GraphKey{Btot, T, want_f, want_j}, text != nullptr, &exec)) return rc;
    HIP_TRY(e, hipGraphLaunch(exec, stream));
    replayed = true;
  }
#endif
  if (!replayed) {
    if (int rc = enqueue_sample(e, stream, text, e->lat_in, Btot, T, nullptr, want_f ? e->feats_int : nullptr, want_j ? e->joints_int : nullptr)) return rc;
  }
  {
    ScatterArgs sa;
    int o = 0, bmax = 0;
    for (int i = 0; i < kMaxRequests; ++i) { sa.lat_out[i] = nullptr; sa.feats_out[i] = nullptr; sa.joints_out[i] = nullptr; sa.off[i] = 0; sa.nb[i] = 0; sa.tmax[i] = 0; }
    for (int i = 0; i < nreq; ++i) {
      sa.lat_out[i] = rq[i].latents_out_dev; sa.feats_out[i] = rq[i].feats_out_dev; sa.joints_out[i] = rq[i].joints_out_dev;
      sa.off[i] = o; sa.nb[i] = rq[i].B; sa.tmax[i] = tmax[i];
      o += rq[i].B; bmax = std::max(bmax, (int)rq[i].B);
    }
    sa.lat = e->lat; sa.feats = e->feats_int; sa.joints = e->joints_int; sa.T = T; sa.D = (int)D; sa.NF = (int)NF; sa.NJ = (int)NJ;
    MLD_LAUNCH(scatter_results_kernel, dim3((unsigned)nreq, (unsigned)bmax), dim3(256), 0, stream, sa);
    if (check_launch(cio, "scatter_results")) return cio.rc;
  }
  return MLDHIP_OK;
}
}  // namespace

extern "C" {

int mldhip_sample_many(mldhip_handle* e, const mldhip_request* reqs, int32_t nreq, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
  if (is_novae(e)) return e->fail(MLDHIP_ESTATE, "mldhip_sample_many serves the latent models (text or action condition)");
  if (!reqs || nreq < 1 || nreq > 64) return e->fail(MLDHIP_EINVAL, "1..64 requests expected");
  return sample_many_impl(e, reqs, nreq, (hipStream_t)stream_);
}

int mldhip_sample(mldhip_handle* e, const float* text_emb_dev, const float* init_latents_dev, const int32_t* lengths_host,
                  int32_t B, float* latents_out_dev, float* feats_out_dev, float* joints_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
  if (is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the action condition: use mldhip_sample_action");
  if (is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was created for the diffusion-only variant: use mldhip_sample_novae");
  if (joints_out_dev && is_actor(e)) return e->fail(MLDHIP_ESTATE, "joints of the ActorVae feature layout need SMPL (out of scope)");
  if (!text_emb_dev) return e->fail(MLDHIP_EINVAL, "null input pointer");
  return sample_impl(e, text_emb_dev, nullptr, init_latents_dev, lengths_host, B, latents_out_dev, feats_out_dev, joints_out_dev, stream_);
}

int mldhip_sample_action(mldhip_handle* e, const int32_t* actions_host, const float* init_latents_dev, const int32_t* lengths_host,
                         int32_t B, float* latents_out_dev, float* feats_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  const DenView v = den_view(e, R);
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
  DeviceGuard dg(e->device);
  if (is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the action condition: use mldhip_denoiser_forward_action");
  if (is_novae(e)) return e->fail(MLDHIP_ESTATE, "engine was created for the diffusion-only variant: use mldhip_denoiser_forward_novae");
  if (!text_emb_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  return denoiser_forward_impl(e, sample_dev, timestep, text_emb_dev, nullptr, R, out_dev, stream_);
}

int mldhip_denoiser_forward_action(mldhip_handle* e, const float* sample_dev, int32_t timestep, const int32_t* actions_host,
                                   int32_t R, float* out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
  if (!is_action(e)) return e->fail(MLDHIP_ESTATE, "engine was created with the text condition: use mldhip_denoiser_forward");
  if (!actions_host) return e->fail(MLDHIP_EINVAL, "null pointer");
  return denoiser_forward_impl(e, sample_dev, timestep, nullptr, actions_host, R, out_dev, stream_);
}

int mldhip_sample_novae(mldhip_handle* e, const float* text_emb_dev, const float* init_latents_dev, const int32_t* lengths_host,
                        int32_t B, const float* step_noise_dev, uint64_t seed, float* feats_out_dev, float* joints_out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  if (int rc = novae_prologue(e, stream, text_emb_dev, init_latents_dev, B, T)) return rc;
  const int n = e->cfg.num_inference_steps;
#if !defined(MLDHIP_SIM)
  if (e->cfg.use_graph && !step_noise_dev) {
    // ~114 launches per step: a 1000-step call is 114 k launches.  Issued eagerly they keep one host thread busy for the
    // whole call, so a second call on another stream cannot even be enqueued before the first is nearly done.  The steps
    // are therefore captured once per (B, Tmax) in chunks of kStepChunk and replayed; everything a step needs is a
    // constant of (weights, step index) except the Philox seed, which the step kernel reads from the workspace.
    e->ctxs[e->cur_ctx].seed_host = seed;
    HIP_TRY(e, hipMemcpyAsync(e->seed_slot, &e->ctxs[e->cur_ctx].seed_host, sizeof seed, hipMemcpyHostToDevice, stream));
    auto& graphs = e->ctxs[e->cur_ctx].step_graphs;
    const int nchunks = (n + kStepChunk - 1) / kStepChunk;
    if (graphs.size() + nchunks > 512) {
      drain_context(e->ctxs[e->cur_ctx]);
      for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
      graphs.clear();
    }
    for (int ch = 0; ch < nchunks; ++ch) {
      auto key = std::make_tuple((int)B, T, ch);
      auto it = graphs.find(key);
      if (it == graphs.end()) {
        hipGraph_t graph = nullptr;
        HIP_TRY(e, hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
        int rc = novae_steps(e, e->cap_stream, B, T, ch * kStepChunk, std::min(n, (ch + 1) * kStepChunk), nullptr, 0,
                             reinterpret_cast<const unsigned long long*>(e->seed_slot));
        hipError_t st = hipStreamEndCapture(e->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "hipStreamEndCapture(steps): %s", hipGetErrorString(st));
        hipGraphExec_t exec = nullptr;
        st = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "hipGraphInstantiate(steps): %s", hipGetErrorString(st));
        it = graphs.emplace(key, exec).first;
      }
      HIP_TRY(e, hipGraphLaunch(it->second, stream));
    }
    return novae_epilogue(e, stream, B, T, feats_out_dev, joints_out_dev);
  }
#endif
  if (int rc = novae_steps(e, stream, B, T, 0, n, step_noise_dev, seed, nullptr)) return rc;
  return novae_epilogue(e, stream, B, T, feats_out_dev, joints_out_dev);
}

int mldhip_denoiser_forward_novae(mldhip_handle* e, const float* sample_dev, int32_t timestep, const float* text_emb_dev,
                                  const int32_t* lengths_host, int32_t R, int32_t T, float* out_dev, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  novae_fold_memory(c, e->TKV_one, 1, (long long)2 * D, e->TKW_one, e->TKU_one, e->TKC_one);
  novae_text_memory(c, text_emb_dev, R);
  novae_pad_input(c, sample_dev, (long long)R * T, 1);
  novae_denoiser_body(c, R, T, e->TKV_one, (long long)2 * D, NovaeFold{e->TKW_one, e->TKU_one, e->TKC_one, 1}, out_dev);
  return c.rc;
}

int mldhip_ddpm_step(mldhip_handle* e, const float* eps_dev, int32_t timestep, const float* sample_dev, const float* noise_dev,
                     uint64_t seed, int32_t step_index, float* prev_dev, int64_t n, void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  DeviceGuard dg(e->device);
  if (!out_dev || n < 1) return e->fail(MLDHIP_EINVAL, "null pointer / n < 1");
  Ctx c{e, (hipStream_t)stream_};
  MLD_LAUNCH(philox_normal_kernel, dim3((unsigned)std::min<long long>(4096, (n / 4 + 256) / 256)), dim3(256), 0, c.stream, out_dev, (long long)n,
             (unsigned long long)seed, (unsigned)step_index);
  return check_launch(c, "philox_normal");
}

int mldhip_vae_decode(mldhip_handle* e, const float* z_dev, const int32_t* lengths_host, int32_t B, float* feats_out_dev,
                      void* stream_) {
  if (!e) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  DeviceGuard dg(e->device);
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
  DeviceGuard dg(e->device);
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
  DeviceGuard dg(e->device);
  if (is_actor(e) || e->cfg.nfeats < 67) return e->fail(MLDHIP_ESTATE, "feats2joints implements the HumanML3D layout only (SMPL-based layouts are out of scope)");
  if (!e->finalized || !e->group_ready[2]) return e->fail(MLDHIP_ESTATE, "feats2joints before finalize / mean,std not loaded");
  if (!feats_dev || !joints_out_dev) return e->fail(MLDHIP_EINVAL, "null pointer");
  if (B < 1 || T < 1 || T > 512) return e->fail(MLDHIP_EINVAL, "B >= 1 and 1 <= T <= 512 required");
  Ctx c{e, (hipStream_t)stream_};
  e->phase = 2;
  joints_body(c, feats_dev, B, T, joints_out_dev);
  return c.rc;
}


// Measurement hooks: only in the hooks build (make hooks -> libmldhip_hooks.so, include/mldhip_hooks.h); the production library exports the sampling surface only.
#if defined(MLDHIP_HOOKS)
int mldhip_profile_kernel(mldhip_handle* e, const char* name, int32_t B, int32_t T, int32_t iters, double* flops_per_launch,
                          void* stream_) {
  // Launches ONE kernel of the sampling path `iters` times back-to-back on `stream` at its production
  // shape, on the engine's own buffers (call after a sample() so they hold real activations).  The
  // caller brackets the call with events on the same stream (bench.py does) -- no timing happens here.
  if (!e || !name || !flops_per_launch) return MLDHIP_EINVAL;
  DeviceGuard dg(e->device);
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
  const DenView v = den_view(e, R);
  for (int it = 0; it < iters && !c.rc; ++it) {
    if (n == "den_qkv") {            // with the LN2-on-load prologue of a typical layer (sums the 4 FFN2 slabs)
      den_qkv(c, v, DL, den_layer_output(e, v, e->den[mid - 1], v.S[mid - 1]));
      *flops_per_launch = 2.0 * M * D * 3 * D;
    } else if (n == "den_outproj") {
      den_outproj(c, v, DL);
      *flops_per_launch = 2.0 * M * D * D + 4.0 * M * 3 * D;
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
    } else if (n == "dec_ffn") {      // the whole feed-forward block as the layer runs it (one fused launch in the split-bf16 modes)
      ffn_block(c, e->H1, e->Hb, (int)M, e->dec[mid].l1_w, e->dec[mid].l1_b, e->dec[mid].l2_w, e->dec[mid].l2_b, e->dec[mid].n3_w, e->dec[mid].n3_b, 0);
      *flops_per_launch = 4.0 * M * D * F;
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
  DeviceGuard dg(e->device);
  constexpr int64_t kMax = 512 * 8 * 8;
  if (!e->trace_buf && hipMalloc((void**)&e->trace_buf, kMax * sizeof(uint64_t)) != hipSuccess) return e->fail(MLDHIP_EHIP, "hipMalloc(trace)");
  if (std::string(name) == "den_loop_phases") {
    // the persistent loop's own phase counters ("fused_dbg" 5): written by the last sample call, 16 values per wave of the first 64 workgroups
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_));
    const int64_t n = std::min<int64_t>(cap_u64, 64 * 8 * 16);      // 16 counters per wave: two 8-value records
    HIP_TRY(e, hipMemcpy(out_host, e->trace_buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)(n / 64);
  }
  if (std::string(name) == "den_cluster_status") {
    // the cluster loop's status words of the last call: [0] timeout flag, [1] clusters that span XCDs, then per workgroup (first 256) its XCC id + 1 (debug builds of the kernel)
    if (!e->cl_flags) return e->fail(MLDHIP_ESTATE, "no cluster loop on this handle");
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_));
    const size_t ncl = std::min<size_t>(kClMaxClusters, (e->cfg.max_batch + 7) / 8);
    const int64_t n = std::min<int64_t>(cap_u64, 8);
    HIP_TRY(e, hipMemcpy(out_host, reinterpret_cast<unsigned*>(e->cl_flags) + ncl * kClFlagWords, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 1;
  }
  if (std::string(name) == "den_cluster_xbuf") {
    // the exchange region of cluster 0 as the last cluster-loop call left it (kernels/loop_cluster.hpp: AO, h1, Y, Z, H of the last two layers) + its flags:
    // what tests compare between the simulator and the GPU
    if (!e->cl_xbuf) return e->fail(MLDHIP_ESTATE, "no cluster loop on this handle");
    HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_));
    const int64_t n = std::min<int64_t>(cap_u64, (int64_t)kClXFloats / 2);
    const int cl = std::max(0, std::min<int>(B, (int)std::min<size_t>(kClMaxClusters, (e->cfg.max_batch + 7) / 8) - 1));      // B = cluster index here
    HIP_TRY(e, hipMemcpy(out_host, e->cl_xbuf + (size_t)cl * kClXFloats, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return (int)(n / 64);
  }
  double fl = 0;
  if (int rc = mldhip_profile_kernel(e, name, B, T, 3, &fl, stream_)) return rc;
  // the traced build is its own kernel: two launches of it first (code fetched, instruction cache warm), then the one that is kept
  e->trace_on = e->trace_buf;
  int rc = mldhip_profile_kernel(e, name, B, T, 2, &fl, stream_);
  if (!rc && hipMemsetAsync(e->trace_buf, 0, kMax * sizeof(uint64_t), (hipStream_t)stream_) != hipSuccess) rc = e->fail(MLDHIP_EHIP, "hipMemsetAsync(trace)");
  if (!rc) rc = mldhip_profile_kernel(e, name, B, T, 1, &fl, stream_);
  e->trace_on = nullptr;
  if (rc) return rc;
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream_));
  const int64_t n = std::min<int64_t>(cap_u64, kMax);
  HIP_TRY(e, hipMemcpy(out_host, e->trace_buf, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return (int)(n / 64);
}

#endif  // MLDHIP_HOOKS

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
