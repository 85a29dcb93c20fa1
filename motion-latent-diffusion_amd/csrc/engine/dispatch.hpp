// Launch helpers: which GEMM tile / kernel instance a call maps to.
// Part of libmldhip's single translation unit (included by ../mldhip.hip, in this order: state, params, dispatch,
// path_latent, path_novae).  Internal linkage throughout (anonymous namespace) except the handle type itself.
#pragma once

namespace {

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

// Tile configurations.  Tiny one-off GEMMs (time MLP, text projection, per-sample cross-attention vectors; M up to
// e->small_m rows) use the register-direct 16x64 one-tile-per-wave shape; everything else streams both panels through LDS
// (64x128 tiles on 8 waves, 64x256 with the LayerNorm epilogue; 32x64 on 4 waves for the loop's K = 1024 GEMM at large M).

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
      if constexpr (!LN && PREC == PREC_F32) { MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 12>), grid, dim3(WM * WN * 64), lds, c.stream, a); }
      else c.rc = c.e->fail(MLDHIP_EINVAL, "staged GEMM: K=384 is built for the plain fp32 tile only");
      break;
    case 16: MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 16>), grid, dim3(WM * WN * 64), lds, c.stream, a); break;
    case 32: MLD_LAUNCH((gemm_kernel<WM, WN, MREP, NREP, LN, true, PREC, 32>), grid, dim3(WM * WN * 64), lds, c.stream, a); break;
    default: c.rc = c.e->fail(MLDHIP_EINVAL, "staged GEMM: K=%d not in {256,384,512,1024}", a.K1 + a.K2);
  }
}
// Operand format of the LDS-staged GEMMs outside the latent reverse loop (mldhip.h MLDHIP_PREC_*): the decoder / encoder
// GEMMs (phase 1) and every GEMM of the diffusion-only variant.
int staged_prec(const E* e) {
  const bool big = e->phase == 1 || e->cfg.vae_arch == MLDHIP_VAE_NONE;
  switch (e->cfg.precision) {
    case MLDHIP_PREC_BF16X3_DECODE: return (big && e->split_decode_ok) ? PREC_BF16X3 : PREC_F32;   // split_decode_ok: finalize's range probe
    case MLDHIP_PREC_BF16: return PREC_BF16;
    default: return PREC_F32;
  }
}
// Operand format of the latent reverse loop's GEMMs (tile32.hpp / strip.hpp / the 32x64 staged FFN2)
int loop_prec(const E* e) {
  return e->cfg.precision == MLDHIP_PREC_BF16 ? PREC_BF16 : PREC_F32;
}

// Operand format of the LATENCY family (tile32.hpp: one bs-64 batch at a time): the split-f16 mode runs them on split-f16 MFMAs too
// (option "tile_x3", on by default) -- 24 matrix instructions of 16 cycles per wave instead of 64 of 32, same 22-bit products as the
// persistent loop of that mode
int latency_prec(const E* e) {
  return (e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE && e->tile_x3 && e->split_loop_ok) ? PREC_BF16X3 : loop_prec(e);
}
// ... and the persistent loop (loop_fused.hpp) streams the split-f16 image of its weights
bool fused_split(const E* e) { return e->loop_stream_x3 && e->fused_x3 && e->split_loop_ok; }

// split-bf16 mode: read W from the pre-split image of the weight arena when it lives there (derived tables in a workspace do not)
void use_split_weights(const E* e, GemmArgs& a, int prec) {
  // the image holds one (hi | lo) record per ALIGNED group of 32 floats of the arena: a weight view that does not start on a
  // group boundary, or whose rows / K slices do not, keeps the in-kernel split
  if (prec == PREC_BF16X3 && e->arena_x3 && a.W >= e->arena && a.W < e->arena + e->arena_floats &&
      (a.W - e->arena) % 32 == 0 && a.ldw % 32 == 0 && a.sW % 32 == 0) {
    a.W = e->arena_x3 + (a.W - e->arena);
    a.w_split = 1;
  }
}

// The MFMA-bound GEMMs of the diffusion-only variant (d = 512: K in {512, 1024}, N in {512, 1024, 1536}, M = 25 088 rows at the BASELINE
// shape) on the software-pipelined 128 x 256 tile (kernels/gemm_pipe.hpp; option "gemm_pipe"): split-f16 operands, W from the pre-split
// image, one K segment, N a multiple of the tile width, no per-row masks.  Everything else keeps the 64 x 128 tile.
bool use_gemm_pipe(const E* e, const GemmArgs& a, int prec, int nz) {
  const int K = a.K1 + a.K2;
  return e->gemm_pipe && e->cfg.vae_arch == MLDHIP_VAE_NONE && prec == PREC_BF16X3 && a.w_split && nz == 1 && a.K2 == 0 && (K == 512 || K == 1024) &&
         a.N % 256 == 0 && (e->gemm_pipe == 2 || a.M >= e->gemm_pipe_min_rows) && !a.lens && !a.skip_lens && !a.relu_in && !e->trace_on && (a.lda & 3) == 0 && (a.ldy & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.bias)) & 15) == 0;   // 16-byte row pieces, bias quads, tile stores
}

void gemm(Ctx& c, const GemmArgs& a_, int nz = 1) {
  GemmArgs a = a_;
  const int K = a.K1 + a.K2;
  const bool small = a.M <= c.e->small_m || (K != 256 && K != 384 && K != 512 && K != 1024);
  const int prec = K == 384 ? PREC_F32 : staged_prec(c.e);     // K = 384 (padded 263-wide features): fp32 tile only
  if (!small) use_split_weights(c.e, a, prec);
  if (!small && use_gemm_pipe(c.e, a, prec, nz)) {
    const dim3 grid(gemm_pipe_grid<2, 4, 4, 4>(a.M, a.N));
    constexpr int lds = gemm_pipe_lds_bytes<2, 4, 4, 4>();
    if (K == 512) { MLD_LAUNCH((gemm_pipe_x3_kernel<2, 4, 4, 4, 16, 2>), grid, dim3(512), lds, c.stream, a); }
    else { MLD_LAUNCH((gemm_pipe_x3_kernel<2, 4, 4, 4, 32, 2>), grid, dim3(512), lds, c.stream, a); }
  } else if (small) {
    dim3 grid((a.M + 15) / 16, (a.N + 63) / 64, nz);
    MLD_LAUNCH((gemm_kernel<1, 4, 1, 1, false>), grid, dim3(256), 0, c.stream, a);
  } else {
    dim3 grid((a.M + 63) / 64, (a.N + 127) / 128, nz);         // 64x128 tile on 8 waves (2 per SIMD)
    if (prec == PREC_BF16X3) launch_staged<2, 4, 2, 2, false, PREC_BF16X3>(c, a, grid);
    else if (prec == PREC_BF16) launch_staged<2, 4, 2, 2, false, PREC_BF16>(c, a, grid);
    else launch_staged<2, 4, 2, 2, false, PREC_F32>(c, a, grid);
  }
  count(c);
  check_launch(c, "gemm");
}

// 32x64 tiles on 4 waves (27.6 KB of LDS: several workgroups per CU): the loop's FFN2 at M >= 768, where N = 256 gives
// only M/64 x 2 of the big tiles.  K in {256, 512, 1024}; fp32.
void gemm_tile_32x64(Ctx& c, const GemmArgs& a, int prec, int nz = 1) {
  const dim3 grid((a.M + 31) / 32, (a.N + 63) / 64, nz);
  if (prec == PREC_BF16) launch_staged<2, 2, 1, 2, false, PREC_BF16>(c, a, grid);
  else launch_staged<2, 2, 1, 2, false, PREC_F32>(c, a, grid);
  count(c);
  check_launch(c, "gemm_32x64");
}

void gemm_ln(Ctx& c, const GemmArgs& a_) {   // N == 256; full rows per workgroup (64 x 256 tile on 8 waves)
  GemmArgs a = a_;
  const int prec = staged_prec(c.e);
  use_split_weights(c.e, a, prec);
  const dim3 grid((a.M + 63) / 64, 1, 1);
  if (prec == PREC_BF16X3) launch_staged<2, 4, 2, 4, true, PREC_BF16X3>(c, a, grid);
  else if (prec == PREC_BF16) launch_staged<2, 4, 2, 4, true, PREC_BF16>(c, a, grid);
  else launch_staged<2, 4, 2, 4, true, PREC_F32>(c, a, grid);
  count(c);
  check_launch(c, "gemm_ln");
}

GemmArgs lin_args(const float* A, int lda, int K, const float* W, const float* b, float* Y, int ldy, int M, int N) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.K1 = K; g.W = W; g.ldw = K; g.bias = b; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N;
  return g;
}

}  // namespace
