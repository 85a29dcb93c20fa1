// Latent-diffusion paths (configs 1-3, 5): denoiser loop, VAE decode / encode, feats2joints, sample() enqueue.
// Part of libmldhip's single translation unit (included by ../mldhip.hip, in this order: state, params, dispatch,
// path_latent, path_novae).  Internal linkage throughout (anonymous namespace) except the handle type itself.
#pragma once

namespace {

// ---- denoiser layer pipeline (4 launches per encoder layer) on one of two kernel families -------------------------
//   latency    (kernels/tile32.hpp): everything loaded before the first MFMA, split-K slabs summed by the consumer;
//              M = 6B <= a few hundred rows (one bs-64 request: 384).
//   throughput (kernels/strip.hpp + the 32x64 staged GEMM): A strip resident, weights streamed, 3 workgroups per CU,
//              no split-K (one raw slab per GEMM); M >= strip_min_rows (several requests coalesced into one chain).
// Both use the same data flow: a GEMM with K > 256 or a following LayerNorm leaves RAW fp32 partial slabs, and the
// consumer's A prologue applies slab sum + bias + residual + LayerNorm (or the 3-token attention).

void tile32(Ctx& c, const Tile32Args& a_, int nz) {
  Tile32Args a = a_;
  a.trace = c.e->trace_on;
  // 16-row K-split tiles for the narrow (N = 256) GEMMs: more workgroups, fewer bytes and MFMAs per CU
  const bool mt16 = a.N <= 256 && ((a.M + 15) / 16) * ((a.N + 63) / 64) * nz <= 256;
  const int mt = mt16 ? 16 : 32;
  dim3 grid((a.M + mt - 1) / mt, (a.N + 63) / 64, nz);
  const int ns = a.src[0].attn_R > 0 ? 0 : a.src[0].nsplit;
  const int prec = latency_prec(c.e);
  if (prec == PREC_BF16X3 && c.e->arena_x3 && a.W >= c.e->arena && a.W < c.e->arena + c.e->arena_floats &&
      (a.W - c.e->arena) % 32 == 0 && a.ldw % 32 == 0) {
    a.W = c.e->arena_x3 + (a.W - c.e->arena);
    a.w_split = 1;
  }
  const bool attn = a.src[0].attn_R > 0, two = ns > 0 && nz > a.nz0;
#define MLD_T32P(MT, NS, MODE)                                                                                   \
  do {                                                                                                           \
    if (a.trace && prec == PREC_BF16X3) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, true, PREC_BF16X3, MODE>), grid, dim3(512), kT32LdsBytes, c.stream, a); } \
    else if (a.trace) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, true, PREC_F32, MODE>), grid, dim3(512), kT32LdsBytes, c.stream, a); }  \
    else if (prec == PREC_BF16) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, false, PREC_BF16, MODE>), grid, dim3(512), kT32LdsBytes, c.stream, a); } \
    else if (prec == PREC_BF16X3) { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, false, PREC_BF16X3, MODE>), grid, dim3(512), kT32LdsBytes, c.stream, a); } \
    else { MLD_LAUNCH((gemm_tile32_kernel<MT, NS, false, PREC_F32, MODE>), grid, dim3(512), kT32LdsBytes, c.stream, a); }              \
  } while (0)
#define MLD_T32(MT, NS)                                                                                          \
  do { if ((NS) == 0 ? attn : two) MLD_T32P(MT, NS, 1); else MLD_T32P(MT, NS, 0); } while (0)
#define MLD_T32_NS(MT)                                                                                           \
  switch (ns) {                                                                                                  \
    case 0: MLD_T32(MT, 0); break;                                                                               \
    case 1: MLD_T32(MT, 1); break;                                                                               \
    case 2: MLD_T32(MT, 2); break;                                                                               \
    case 4: MLD_T32(MT, 4); break;                                                                               \
    default: c.rc = c.e->fail(MLDHIP_EINVAL, "tile32: unsupported slab count %d", ns); return;                   \
  }
  if (mt16) { MLD_T32_NS(16) } else { MLD_T32_NS(32) }
#undef MLD_T32_NS
#undef MLD_T32
#undef MLD_T32P
  count(c);
  check_launch(c, "gemm_tile32");
}

// throughput family: K = 256 (one source) or 512 (skip linear: src[0] | src[1]); src[0] plain, 1- or 2-slab combine, or attention.
// Wide GEMMs (N a multiple of 128, N >= 512: QKV, FFN1) take 32 x 128 tiles: half as many workgroups repeat one A prologue.
// (QKV alone is faster on 32 x 64 tiles at 1 920 rows -- 13.5 vs 14.9 us, one resident round of 720 workgroups -- but with four
// calls in flight the end-to-end rate is 2 % LOWER: 12.77 vs 13.04 k motions/s, profiles/r02_strip_options_ab.json.)
void strip(Ctx& c, const Tile32Args& a_, int nsrc) {
  Tile32Args a = a_;
  a.trace = c.e->trace_on;
  const bool attn = a.src[0].attn_R > 0;
  const int ns = attn ? 0 : a.src[0].nsplit;
  const bool wide = !attn && nsrc == 1 && a.N % 128 == 0 && a.N >= 512;      // (round 2's "strip_wide" / "strip_waves" / "strip_ffn2_split" knobs were retired in round 6: the settled forms are how it works)
  const dim3 grid((a.M + 31) / 32, wide ? a.N / 128 : (a.N + 63) / 64, 1);
  const int prec = loop_prec(c.e);
#define MLD_STRIP(NS, NSRC, ATTN, ACT, CT, NW)                                                                                     \
  do {                                                                                                                             \
    if (prec == PREC_BF16) { MLD_LAUNCH((gemm_strip_kernel<NS, NSRC, ATTN, PREC_BF16, ACT, CT, NW>), grid, dim3(64 * NW), (strip_lds_bytes<NSRC, CT>()), c.stream, a); } \
    else { MLD_LAUNCH((gemm_strip_kernel<NS, NSRC, ATTN, PREC_F32, ACT, CT, NW>), grid, dim3(64 * NW), (strip_lds_bytes<NSRC, CT>()), c.stream, a); }       \
  } while (0)
  // 8 waves per workgroup (one 16-row tile per wave)
#define MLD_STRIP_W(NS, NSRC, ATTN, ACT)                                            \
  do {                                                                              \
    if (wide) MLD_STRIP(NS, NSRC, ATTN, ACT, 2, 8);                                 \
    else MLD_STRIP(NS, NSRC, ATTN, ACT, 1, 8);                                      \
  } while (0)
  if (a.trace) {                              // measurement builds (mldhip_profile_trace): the fp32 8-wave kernels of the encoder layer
    if (prec != PREC_F32 || nsrc != 1) { c.rc = c.e->fail(MLDHIP_EINVAL, "strip: traces exist for the fp32 8-wave layer kernels only"); return; }
    if (attn) { MLD_LAUNCH((gemm_strip_kernel<0, 1, true, PREC_F32, 0, 1, 8, true>), grid, dim3(512), (strip_lds_bytes<1, 1>()), c.stream, a); }
    else if (wide && ns == 1 && a.act == 1) { MLD_LAUNCH((gemm_strip_kernel<1, 1, false, PREC_F32, 1, 2, 8, true>), grid, dim3(512), (strip_lds_bytes<1, 2>()), c.stream, a); }
    else if (wide && ns == 2 && a.act == 0) { MLD_LAUNCH((gemm_strip_kernel<2, 1, false, PREC_F32, 0, 2, 8, true>), grid, dim3(512), (strip_lds_bytes<1, 2>()), c.stream, a); }
    else { c.rc = c.e->fail(MLDHIP_EINVAL, "strip: no traced build of this shape"); return; }
    count(c);
    check_launch(c, "gemm_strip(trace)");
    return;
  }
  if (a.act != 0 && !(a.act == 1 && ns == 1 && nsrc == 1 && !attn)) { c.rc = c.e->fail(MLDHIP_EINVAL, "strip: activation %d is built for the FFN1 shape only", a.act); return; }
  if (attn && nsrc == 1) MLD_STRIP(0, 1, true, 0, 1, 8);
  else if (ns == 0 && nsrc == 1) MLD_STRIP_W(0, 1, false, 0);
  else if (ns == 1 && nsrc == 1 && a.act == 1) MLD_STRIP_W(1, 1, false, 1);
  else if (ns == 1 && nsrc == 1) MLD_STRIP_W(1, 1, false, 0);
  else if (ns == 2 && nsrc == 1) MLD_STRIP_W(2, 1, false, 0);
  else if (ns == 1 && nsrc == 2) MLD_STRIP(1, 2, false, 0, 1, 8);
  else if (ns == 2 && nsrc == 2) MLD_STRIP(2, 2, false, 0, 1, 8);
  else { c.rc = c.e->fail(MLDHIP_EINVAL, "strip: unsupported source (slabs %d, segments %d)", ns, nsrc); return; }
#undef MLD_STRIP_W
#undef MLD_STRIP
  count(c);
  check_launch(c, "gemm_strip");
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

// The denoiser workspace as one chain sees it: rows [0, 3R) of every row-indexed buffer.
struct DenView {
  float *X0, *QKV, *FF, *H1, *Ha, *Po, *Pf, *Ps, *S[8], *lat;
  int R;            // samples in the CFG batch (uncond half first)
  bool strip;       // throughput kernel family (see above)
  int ffn_slabs, skip_slabs;   // raw partial slabs FFN2 / the skip linear leave behind
};

// Which family runs the reverse loop of a call is a measured table (tools/ab_crossover.py -> profiles/r04_loop_crossover.json; ms per loop-only
// call of B motions, MI355X):          B =    64    128    192    256    320    640  | exact fp32:  256    640   1 024  1 280  1 536
//   latency kernels (tile32.hpp)           11.1   15.2   19.0   21.0     --     --  |             27.2     --     --     --     --
//   column-split throughput (strip.hpp)    18.3   18.4   23.2   25.4   26.3   40.7  |             25.4   40.5   59.9   74.4   87.1
//   persistent loop (loop_fused.hpp)       19.8   19.5   19.3   19.2   19.1   18.9  |             73.0   73.2   73.5   73.7   73.8
// Split-f16 mode: the latency kernels (split-f16 MFMAs under "tile_x3") up to 191 motions, the persistent loop from 192 -- the
// column-split family, whose loop arithmetic is fp32 in that mode, never wins there.  Exact fp32: latency kernels below 128 motions
// ("strip_min_rows" 768), column-split up to 1 279, persistent loop from 1 280.
bool use_strip(const E* e, int rows) {
  if (e->loop_kernel == 2) return true;
  if (e->loop_kernel != 0) return false;
  if (latency_prec(e) == PREC_BF16X3 && rows < 6 * 256) return false;      // split-f16 latency kernels beat the fp32 column-split ones wherever both run
  return rows >= e->strip_min_rows;
}

DenView den_view(E* e, int R) {
  DenView v;
  v.X0 = e->X0; v.QKV = e->QKV; v.FF = e->FF; v.H1 = e->H1; v.Ha = e->Ha;
  v.Po = e->Po; v.Pf = e->Pf; v.Ps = e->Ps;
  for (int i = 0; i < 8; ++i) v.S[i] = e->S[i];
  v.lat = e->lat;
  v.R = R;
  v.strip = use_strip(e, 3 * R);
  // throughput kernels: K slices of FFN2 no narrower than 256 (the staged GEMM takes K in {256, 512, 1024})
  v.ffn_slabs = v.strip ? std::min(2, e->cfg.ff_size / 256) : e->cfg.ff_size / 256;
  v.skip_slabs = v.strip ? 1 : 2;
  return v;
}
long long den_slab(const E* e) { return (long long)6 * e->cfg.max_batch * 256; }

// QKV projection; `x` describes how the layer input rows are obtained (and where they are written back).
void den_qkv(Ctx& c, const DenView& v, const EncLayerP& L, const ASrc& x) {
  Tile32Args a;
  a.src[0] = x; a.nz0 = 1; a.W = L.in_w; a.ldw = 256; a.bias = L.in_b; a.Y = v.QKV; a.ldy = 768; a.M = 3 * v.R; a.N = 768;
  if (v.strip) strip(c, a, 1); else tile32(c, a, 1);
}
// out-projection of the 3-token self-attention (computed while the A tile is assembled) -> raw slab Po
void den_outproj(Ctx& c, const DenView& v, const EncLayerP& L) {
  Tile32Args a;
  a.src[0].base = v.QKV; a.src[0].attn_R = v.R;
  a.nz0 = 1; a.W = L.out_w; a.ldw = 256; a.P = v.Po; a.pstride = 0; a.M = 3 * v.R; a.N = 256;
  if (v.strip) strip(c, a, 1); else tile32(c, a, 1);
}
// h1 = LN1(x + out_proj) assembled on load (written to H1), FF = gelu(h1 W1^T + b1)
void den_ffn1(Ctx& c, const DenView& v, const EncLayerP& L, const float* xn) {
  const int F = c.e->cfg.ff_size;
  Tile32Args a;
  a.src[0] = combine_src(v.Po, 1, 0, L.out_b, xn, L.n1_w, L.n1_b, v.H1);
  a.nz0 = 1; a.W = L.l1_w; a.ldw = 256; a.bias = L.l1_b; a.act = 1; a.Y = v.FF; a.ldy = F; a.M = 3 * v.R; a.N = F;
  if (v.strip) strip(c, a, 1); else tile32(c, a, 1);
}
// FFN2 -> raw slabs Pf (ff_size/256 K-slices on the latency kernels, one full-K slab on the throughput kernels);
// bias, residual and norm2 are applied by whoever reads them
void den_ffn2(Ctx& c, const DenView& v, const EncLayerP& L) {
  const int F = c.e->cfg.ff_size;
  if (v.strip) {
    // `ffn_slabs` K slices (blockIdx.z) -> as many raw slabs: with one slice the N = 256 GEMM has M/32 x 4 workgroups, fewer
    // than CUs at M <= 2 048, each walking all 32 K chunks behind a 4-deep prefetch ring (latency bound: 51 TF measured)
    const int nz = v.ffn_slabs, Kz = F / nz;
    GemmArgs g = lin_args(v.FF, F, Kz, L.l2_w, nullptr, v.Pf, 256, 3 * v.R, 256);
    g.ldw = F; g.sA = Kz; g.sW = Kz; g.sY = den_slab(c.e);
    gemm_tile_32x64(c, g, loop_prec(c.e), nz);
    return;
  }
  Tile32Args a;
  a.src[0] = plain_src(v.FF, F);
  a.nz0 = F / 256; a.W = L.l2_w; a.ldw = F; a.P = v.Pf; a.pstride = den_slab(c.e); a.M = 3 * v.R; a.N = 256;
  tile32(c, a, F / 256);
}
ASrc den_layer_output(E* e, const DenView& v, const EncLayerP& L, float* write_back) {   // LN2(sum Pf + b2 + h1)
  return combine_src(v.Pf, v.ffn_slabs, den_slab(e), L.l2_b, v.H1, L.n2_w, L.n2_b, write_back);
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
    den_ffn1(c, v, P_, xn);
    den_ffn2(c, v, P_);
    if (l + 1 == L) break;
    if (l < nb) {
      // next layer input = LN2(...), kept in S[l] for the skip connection (written by the next QKV prologue)
      x = den_layer_output(e, v, P_, v.S[l]);
      xn = v.S[l];
    } else {
      // Linear(cat[x, skip]) as two K segments (cross_attention.py:56-58): segment 0 assembles x = LN2(...) on load,
      // segment 1 reads the stored skip activation; the bias is added by the next QKV prologue.
      const int i = l - nb;
      Tile32Args a;
      a.src[0] = den_layer_output(e, v, P_, nullptr);
      a.src[1] = plain_src(v.S[nb - 1 - i], 256);
      a.nz0 = 1;
      a.W = P(e, "denoiser.encoder.linear_blocks." + std::to_string(i) + ".weight"); a.ldw = 512;
      a.P = v.Ps; a.pstride = den_slab(e); a.M = 3 * v.R; a.N = 256;
      if (v.strip) strip(c, a, 2); else tile32(c, a, 2);
      x = combine_src(v.Ps, v.skip_slabs, den_slab(e), P(e, "denoiser.encoder.linear_blocks." + std::to_string(i) + ".bias"), nullptr,
                      nullptr, nullptr, v.Ha);
      xn = v.Ha;
    }
  }
}

// ---- sample-major persistent loop (kernels/loop_fused.hpp): built for the configurations the released checkpoints use
bool fused_built(const E* e) {
  return !is_novae(e) && e->cfg.latent_dim == 256 && e->cfg.ff_size == 1024 && e->cfg.num_heads == 4 && loop_prec(e) == PREC_F32;
}
bool use_fused(const E* e, int B) {
  // auto: the persistent loop takes the same time for any batch up to 8 x #CUs motions -- 19 ms on split-f16 MFMAs, 73 ms on exact-fp32
  // ones (r04) -- see the measured table at use_strip above: cross-over by operand format
  const int auto_min = e->fused_min_batch > 0 ? e->fused_min_batch : (fused_split(e) ? 192 : 1280);
  return e->loop_ips > 0 && (e->loop_kernel == 3 || (e->loop_kernel == 0 && B >= auto_min));
}

// finalize-time: the denoiser's GEMM weights as the item stream the loop kernel consumes, its small parameters packed, the
// DDIM coefficients of the scheduler's steps.  Item order = the kernel's phase order (loop_fused.hpp).
int build_loop_stream(Ctx& c) {
  E* e = c.e;
  e->loop_ips = 0;
  if (!fused_built(e) || !e->group_ready[0]) return 0;
  const int L = e->cfg.num_layers, nb = (L - 1) / 2, n = e->cfg.num_inference_steps, F = e->cfg.ff_size;
  std::vector<LoopItem> items;
  auto push = [&](const float* w, int ld, int row0, int k0) { items.push_back(LoopItem{(long long)(w - e->arena) + (long long)row0 * ld + k0, ld, 0}); };
  // chunk-major inside a group: the items that multiply the same 32 columns of A are adjacent (loop_fused.hpp run2 / run3)
  for (int l = 0; l < L; ++l) {
    const EncLayerP& P_ = e->den[l];
    for (int hp = 0; hp < 2; ++hp)
      for (int kc = 0; kc < 8; ++kc)
        for (int part = 0; part < 3; ++part) push(P_.in_w, 256, part * 256 + hp * 128, kc * 32);
    for (int kc = 0; kc < 8; ++kc)
      for (int cb = 0; cb < 2; ++cb) push(P_.out_w, 256, cb * 128, kc * 32);
    // feed-forward, in the order of the software pipeline: linear1 of block 0, then [linear1 of block hb + 1, linear2's share of block hb]
    auto f1 = [&](int hb) { for (int kc = 0; kc < 8; ++kc) push(P_.l1_w, 256, hb * 128, kc * 32); };
    auto f2 = [&](int hb) { for (int kc = 0; kc < 4; ++kc) for (int cb = 0; cb < 2; ++cb) push(P_.l2_w, F, cb * 128, hb * 128 + kc * 32); };
    f1(0);
    for (int hb = 0; hb < 8; ++hb) {
      if (hb < 7) f1(hb + 1);
      f2(hb);
    }
    if (l >= nb && l + 1 < L) {
      const float* w = P(e, "denoiser.encoder.linear_blocks." + std::to_string(l - nb) + ".weight");
      for (int half = 0; half < 2; ++half)
        for (int kc = 0; kc < 8; ++kc)
          for (int cb = 0; cb < 2; ++cb) push(w, 512, cb * 128, half * 256 + kc * 32);
    }
  }
  const size_t ips = items.size();
  for (int j = 0; j < 8; ++j) items.push_back(items[j]);      // the ring's look-ahead across the end of a step (loop_fused.hpp gload)
  const size_t nit = items.size(), small_floats = (size_t)L * kLsLayer + (size_t)nb * 256 + 768, tail = (size_t)n * 4;
  if (e->loop_stream) { (void)hipFree(e->loop_stream); e->loop_stream = nullptr; }
  if (e->loop_small) { (void)hipFree(e->loop_small); e->loop_small = nullptr; }
  if (e->loop_stream_x3) { (void)hipFree(e->loop_stream_x3); e->loop_stream_x3 = nullptr; }
  const bool want_x3 = e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE;      // the split mode: a second image of the stream
  LoopItem* items_dev = nullptr;
  if (hipMalloc((void**)&e->loop_stream, nit * kLoopItemFloats * sizeof(float)) != hipSuccess ||
      (want_x3 && hipMalloc((void**)&e->loop_stream_x3, nit * kLoopItemFloats * sizeof(float)) != hipSuccess) ||
      hipMalloc((void**)&e->loop_small, (small_floats + tail) * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&items_dev, nit * sizeof(LoopItem)) != hipSuccess)
    return e->fail(MLDHIP_EHIP, "hipMalloc(sample-major loop tables)");
  e->loop_ddim = e->loop_small + small_floats;
  hipError_t st = hipMemcpy(items_dev, items.data(), nit * sizeof(LoopItem), hipMemcpyHostToDevice);
  if (st == hipSuccess) {
    MLD_LAUNCH(pack_loop_stream_kernel<false>, dim3((unsigned)nit), dim3(512), 0, c.stream, (const float*)e->arena, (const LoopItem*)items_dev, e->loop_stream);
    if (want_x3) MLD_LAUNCH(pack_loop_stream_kernel<true>, dim3((unsigned)nit), dim3(512), 0, c.stream, (const float*)e->arena, (const LoopItem*)items_dev, e->loop_stream_x3);
    check_launch(c, "pack_loop_stream");
    st = hipStreamSynchronize(c.stream);
  }
  (void)hipFree(items_dev);
  if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "sample-major loop tables: %s", hipGetErrorString(st));
  if (c.rc) return c.rc;
  auto put = [&](size_t off, const float* src, size_t nfl) {
    if (st == hipSuccess) st = hipMemcpy(e->loop_small + off, src, nfl * sizeof(float), hipMemcpyDeviceToDevice);
  };
  for (int l = 0; l < L; ++l) {
    const EncLayerP& P_ = e->den[l];
    const size_t o = (size_t)l * kLsLayer;
    put(o + kLsInB, P_.in_b, 768); put(o + kLsOutB, P_.out_b, 256); put(o + kLsN1W, P_.n1_w, 256); put(o + kLsN1B, P_.n1_b, 256);
    put(o + kLsL1B, P_.l1_b, 1024); put(o + kLsL2B, P_.l2_b, 256); put(o + kLsN2W, P_.n2_w, 256); put(o + kLsN2B, P_.n2_b, 256);
  }
  size_t o = (size_t)L * kLsLayer;
  for (int i = 0; i < nb; ++i, o += 256) put(o, P(e, "denoiser.encoder.linear_blocks." + std::to_string(i) + ".bias"), 256);
  put(o, P(e, "denoiser.encoder.norm.weight"), 256);
  put(o + 256, P(e, "denoiser.encoder.norm.bias"), 256);
  put(o + 512, P(e, "denoiser.query_pos.pe"), 256);
  std::vector<float> coef((size_t)n * 4);
  for (int s = 0; s < n; ++s) {
    const DdimCoef k = ddim_coef(e, e->timesteps[s]);
    coef[4 * s] = k.sqrt_at; coef[4 * s + 1] = k.sqrt_1mat; coef[4 * s + 2] = k.sqrt_ap; coef[4 * s + 3] = k.sqrt_1map;
  }
  if (st == hipSuccess) st = hipMemcpy(e->loop_ddim, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice);
  if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "sample-major loop tables: %s", hipGetErrorString(st));
  e->loop_ips = (int)ips;
  return 0;
}

// ---- cluster loop (kernels/loop_cluster.hpp): one bs-64 request (up to 8 x kClMaxClusters motions) as ONE launch of 12-workgroup clusters
int cluster_groups(const E* e, int B);
constexpr int kCusPerXcd = 32;      // MI355X: 8 XCDs x 32 CUs; partitions (CPX / DPX / QPX) expose whole XCDs
bool use_cluster(const E* e, int B) {
  if (!e->cl_stream || !fused_split(e) || e->cluster_failed || e->cluster_foreign || B > kClMaxCall || B > e->cfg.max_batch) return false;
  // every workgroup of a launch needs a CU of its own (125 KB of LDS each) at the same time: the biggest launch of the call against the device's CUs --
  // in total AND per XCD (advisor r5): workgroups go round the XCDs, so the clusters that share a physical XCD (ceil(clusters / XCDs)) must fit its 32 CUs;
  // a partitioned device (2 XCDs, 64 CUs) with 5 clusters x 12 workgroups would put 36 workgroups on a 32-CU XCD and time out on every call
  const int nm = std::min(B, e->cluster_chunk);
  const int members = 3 * cluster_groups(e, nm), ncl = (nm + 7) / 8;
  if (members * ncl > e->num_cus) return false;
  const int xcds = std::max(1, std::min(8, e->num_cus / kCusPerXcd)), per_xcd = e->num_cus / xcds;
  if (members * ((ncl + xcds - 1) / xcds) > per_xcd) return false;
  return e->loop_kernel == 4 || (e->loop_kernel == 0 && B <= e->cluster_max_batch);
}

// column groups per token of a cluster call: 8 (24 workgroups per cluster: the feed-forward block on twice the CUs) while every cluster still has an XCD's 32 CUs
// to itself (up to 8 clusters = 64 motions), 4 (12 workgroups) above; option "cluster_groups" 4 / 8 forces one (8 only where it fits)
int cluster_groups(const E* e, int B) {
  const int xcds = std::max(1, std::min(8, e->num_cus / kCusPerXcd));
  const bool fits8 = (B + 7) / 8 <= 8 && 24 * ((B + 7) / 8) <= e->num_cus && 24 * (((B + 7) / 8 + xcds - 1) / xcds) <= e->num_cus / xcds;
  if (e->cluster_groups == 4 || !fits8) return 4;
  return 8;
}

// sticky status word [2] of any workspace context: a cluster launch of this handle ran into its wait bound since the last look (synchronous: call it behind a sync); clears it
bool cluster_timed_out(E* e) {
  if (!e->cl_flags) return false;
  const size_t words = (size_t)std::min<size_t>(kClMaxClusters, (e->cfg.max_batch + 7) / 8) * kClFlagWords;
  size_t off = 0;
  bool found = false, hit = false;
  for (auto& cv : e->carve) if (cv.first == &e->cl_flags) { off = cv.second; found = true; }
  if (!found) return false;
  for (auto& x : e->ctxs) {
    unsigned st = 0;
    unsigned* w = x.ws ? reinterpret_cast<unsigned*>(x.ws + off) + words + 2 : nullptr;
    if (w && hipMemcpy(&st, w, sizeof st, hipMemcpyDeviceToHost) == hipSuccess && st != 0u) { hit = true; (void)hipMemset(w, 0, sizeof st); }
  }
  return hit;
}

// finalize-time: per column group and wave, the weight fragments (16 rows x 32 k, split-f16) in the order den_cluster_kernel consumes them;
// needs the packed small parameters / DDIM table of build_loop_stream
int build_cluster_stream(Ctx& c) {
  E* e = c.e;
  if (e->cl_stream) { (void)hipFree(e->cl_stream); e->cl_stream = nullptr; }
  if (!fused_built(e) || !e->group_ready[0] || !e->loop_ips || e->cfg.precision != MLDHIP_PREC_BF16X3_DECODE) return 0;
  const int L = e->cfg.num_layers, nb = (L - 1) / 2, F = e->cfg.ff_size;
  std::vector<ClFrag> frags;
  auto push = [&](const float* w, int ld, int row0, int k0) { frags.push_back(ClFrag{(long long)(w - e->arena) + (long long)row0 * ld + k0, ld, 0}); };
  for (int hc = 0; hc < 4; ++hc)
    for (int w = 0; w < 8; ++w) {
      e->cl_wave_off[hc * 8 + w] = (unsigned)(frags.size() * kClFragFloats);
      const size_t first = frags.size();
      for (int l = 0; l < L; ++l) {
        const EncLayerP& P_ = e->den[l];
        for (int kc = 0; kc < 8; ++kc) {                                           // Ph1: waves 0-3 [Q, K] of head hc, waves 4-7 [V]
          if (w < 4) { push(P_.in_w, 256, 64 * hc + 16 * w, 32 * kc); push(P_.in_w, 256, 256 + 64 * hc + 16 * w, 32 * kc); }
          else push(P_.in_w, 256, 512 + 64 * hc + 16 * (w - 4), 32 * kc);
        }
        for (int kc = 0; kc < 2; ++kc)                                             // out-projection, the head's K slice: columns 32 w + 16 j, k = 64 hc + 32 kc
          for (int j = 0; j < 2; ++j) push(P_.out_w, 256, 32 * w + 16 * j, 64 * hc + 32 * kc);
        for (int kc = 0; kc < 8; ++kc)                                             // linear1: hidden columns 256 hc + 32 w + 16 j
          for (int j = 0; j < 2; ++j) push(P_.l1_w, 256, 256 * hc + 32 * w + 16 * j, 32 * kc);
        for (int kc = 0; kc < 16; ++kc) push(P_.l2_w, F, 64 * hc + 16 * (w & 3), 512 * (w >> 2) + 32 * kc);      // linear2: K half w >> 2
        if (l >= nb && l + 1 < L) {
          const float* ws = P(e, "denoiser.encoder.linear_blocks." + std::to_string(l - nb) + ".weight");
          for (int kc = 0; kc < 8; ++kc) push(ws, 512, 64 * hc + 16 * (w & 3), 256 * (w >> 2) + 32 * kc);         // skip linear: x half / parked half
        }
      }
      for (int j = 0; j < kClRing; ++j) frags.push_back(frags[first + j]);        // look-ahead across the end of a step
    }
  // the wide form (den_cluster_kernel<.., 8>): 8 column groups; groups 0-3 are the heads (the same Ph1 sequence), every group holds an eighth of the feed-forward block
  for (int hc = 0; hc < 8; ++hc)
    for (int w = 0; w < 8; ++w) {
      e->cl_wave_off[32 + hc * 8 + w] = (unsigned)(frags.size() * kClFragFloats);
      const size_t first = frags.size();
      for (int l = 0; l < L; ++l) {
        const EncLayerP& P_ = e->den[l];
        if (hc < 4) {
          for (int kc = 0; kc < 8; ++kc) {
            if (w < 4) { push(P_.in_w, 256, 64 * hc + 16 * w, 32 * kc); push(P_.in_w, 256, 256 + 64 * hc + 16 * w, 32 * kc); }
            else push(P_.in_w, 256, 512 + 64 * hc + 16 * (w - 4), 32 * kc);
          }
          for (int kc = 0; kc < 2; ++kc)
            for (int j = 0; j < 2; ++j) push(P_.out_w, 256, 32 * w + 16 * j, 64 * hc + 32 * kc);
        }
        for (int kc = 0; kc < 8; ++kc) push(P_.l1_w, 256, 128 * hc + 16 * w, 32 * kc);                           // linear1: hidden columns 128 hc + 16 w
        for (int kc = 0; kc < 8; ++kc) push(P_.l2_w, F, 32 * hc + 16 * (w & 1), 256 * (w >> 1) + 32 * kc);        // linear2: tile w & 1, K quarter w >> 1
        if (l >= nb && l + 1 < L) {
          const float* ws = P(e, "denoiser.encoder.linear_blocks." + std::to_string(l - nb) + ".weight");
          for (int kc = 0; kc < 4; ++kc) push(ws, 512, 32 * hc + 16 * (w & 1), 128 * (w >> 1) + 32 * kc);         // skip linear: K quarter w >> 1 (0, 1: x; 2, 3: parked)
        }
      }
      for (int j = 0; j < kClRing; ++j) frags.push_back(frags[first + j]);
    }
  ClFrag* fdev = nullptr;
  if (!e->cl_wave_off_dev && hipMalloc((void**)&e->cl_wave_off_dev, 96 * sizeof(unsigned)) != hipSuccess) return e->fail(MLDHIP_EHIP, "hipMalloc(cluster loop offsets)");
  if (hipMemcpy(e->cl_wave_off_dev, e->cl_wave_off, 96 * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) return e->fail(MLDHIP_EHIP, "cluster loop offsets");
  if (hipMalloc((void**)&e->cl_stream, frags.size() * (size_t)kClFragFloats * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&fdev, frags.size() * sizeof(ClFrag)) != hipSuccess)
    return e->fail(MLDHIP_EHIP, "hipMalloc(cluster loop stream)");
  hipError_t st = hipMemcpy(fdev, frags.data(), frags.size() * sizeof(ClFrag), hipMemcpyHostToDevice);
  if (st == hipSuccess) {
    MLD_LAUNCH(pack_cluster_frags_kernel, dim3((unsigned)frags.size()), dim3(64), 0, c.stream, (const float*)e->arena, (const ClFrag*)fdev, e->cl_stream);
    check_launch(c, "pack_cluster_frags");
    st = hipStreamSynchronize(c.stream);
  }
  (void)hipFree(fdev);
  if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "cluster loop stream: %s", hipGetErrorString(st));
  return c.rc;
}

// finalize-time (split precision modes): linear1 / linear2 of every decoder / encoder layer in the item order of
// kernels/ffn_strip.hpp -- run1(0), then [run1(hb), run2(hb - 1)] for hb = 1..7, then run2(7) -- as split-f16 fragment images
int build_ffn_streams(Ctx& c) {
  E* e = c.e;
  e->ffn_stream_of.clear();
  e->gemm_stream_of.clear();
  e->final_stream = nullptr;
  if (e->ffn_streams) { (void)hipFree(e->ffn_streams); e->ffn_streams = nullptr; }
  const bool split = e->cfg.precision == MLDHIP_PREC_BF16X3_DECODE;
  if (!split || is_novae(e) || e->cfg.latent_dim != 256 || e->cfg.ff_size != 1024) return 0;
  std::vector<std::pair<const float*, const float*>> layers;
  if (e->group_ready[1]) for (auto& L : e->dec) layers.push_back({L.l1_w, L.l2_w});
  if (e->group_ready[3]) for (auto& L : e->venc) layers.push_back({L.l1_w, L.l2_w});
  if (layers.empty()) return 0;
  std::vector<LoopItem> items;
  auto push = [&](const float* w, int ld, int row0, int k0) { items.push_back(LoopItem{(long long)(w - e->arena) + (long long)row0 * ld + k0, ld, 0}); };
  for (auto& lw : layers) {
    auto f1 = [&](int hb) { for (int kc = 0; kc < 8; ++kc) push(lw.first, 256, hb * 128, kc * 32); };
    auto f2 = [&](int hb) { for (int kc = 0; kc < 4; ++kc) for (int cb = 0; cb < 2; ++cb) push(lw.second, 1024, cb * 128, hb * 128 + kc * 32); };
    f1(0);
    for (int hb = 1; hb < 8; ++hb) { f1(hb); f2(hb - 1); }
    f2(7);
  }
  // the row-strip GEMMs (kernels/gemm_strip_x3.hpp): per pair of 128-column blocks, per K segment, per chunk, [block 2p, block 2p + 1]
  const size_t ffn_items = items.size();
  std::vector<std::pair<const float*, size_t>> gemm_first;          // weight -> first item of its stream
  auto gstream = [&](const float* w, int N, int K) {
    gemm_first.push_back({w, items.size()});
    for (int pr = 0; pr < N / 256; ++pr)
      for (int sg = 0; sg < K / 256; ++sg)
        for (int kc = 0; kc < 8; ++kc)
          for (int cb = 0; cb < 2; ++cb) push(w, K, (2 * pr + cb) * 128, sg * 256 + kc * 32);
  };
  const int nbv = (e->cfg.num_layers - 1) / 2;
  if (e->group_ready[1]) {
    for (auto& L : e->dec) { gstream(L.in_w, 768, 256); gstream(L.out_w, 256, 256); }
    if (!is_actor(e)) for (int i = 0; i < nbv; ++i) gstream(P(e, "vae.decoder.linear_blocks." + std::to_string(i) + ".weight"), 256, 512);
  }
  if (e->group_ready[3]) {
    for (auto& L : e->venc) { gstream(L.in_w, 768, 256); gstream(L.out_w, 256, 256); }
    if (!is_actor(e)) for (int i = 0; i < nbv; ++i) gstream(P(e, "vae.encoder.linear_blocks." + std::to_string(i) + ".weight"), 256, 512);
  }
  (void)ffn_items;
  // kernels/final_strip.hpp: vae.final_layer.weight [NF][256], 256 < NF <= 264 (the strip's 48 x NF results are parked in its 48 x 264-word image),
  // zero-padded to three 128-row blocks: per chunk [block 0, 1, 2]
  e->final_stream = nullptr;
  const size_t final_first = items.size();
  const int NFv = e->cfg.nfeats;
  if (e->group_ready[1] && !is_actor(e) && NFv > 256 && NFv <= kFsXs) {
    const float* wf = P(e, "vae.final_layer.weight");
    for (int kc = 0; kc < 8; ++kc)
      for (int blk = 0; blk < 3; ++blk) {
        push(wf, 256, blk * 128, kc * 32);
        items.back().pad = std::min(128, NFv - blk * 128);     // valid rows of the block (pack_stream_rows_kernel zero-fills the rest)
      }
  }
  LoopItem* items_dev = nullptr;
  if (hipMalloc((void**)&e->ffn_streams, items.size() * kLoopItemFloats * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&items_dev, items.size() * sizeof(LoopItem)) != hipSuccess)
    return e->fail(MLDHIP_EHIP, "hipMalloc(feed-forward weight streams)");
  hipError_t st = hipMemcpy(items_dev, items.data(), items.size() * sizeof(LoopItem), hipMemcpyHostToDevice);
  if (st == hipSuccess) {
    MLD_LAUNCH(pack_loop_stream_kernel<true>, dim3((unsigned)final_first), dim3(512), 0, c.stream, (const float*)e->arena, (const LoopItem*)items_dev, e->ffn_streams);
    check_launch(c, "pack_ffn_streams");
    if (items.size() > final_first) {
      MLD_LAUNCH(pack_stream_rows_kernel, dim3((unsigned)(items.size() - final_first)), dim3(512), 0, c.stream, (const float*)e->arena,
                 (const LoopItem*)(items_dev + final_first), e->ffn_streams + final_first * (size_t)kLoopItemFloats);
      check_launch(c, "pack_final_stream");
    }
    st = hipStreamSynchronize(c.stream);
  }
  (void)hipFree(items_dev);
  if (st != hipSuccess) return e->fail(MLDHIP_EHIP, "feed-forward weight streams: %s", hipGetErrorString(st));
  for (size_t i = 0; i < layers.size(); ++i) e->ffn_stream_of[layers[i].first] = e->ffn_streams + i * (size_t)kFfnStripItems * kLoopItemFloats;
  for (auto& gf : gemm_first) e->gemm_stream_of[gf.first] = e->ffn_streams + gf.second * (size_t)kLoopItemFloats;
  if (items.size() > final_first) e->final_stream = e->ffn_streams + final_first * (size_t)kLoopItemFloats;
  return c.rc;
}

FinalArgs den_final_args(E* e, const DenView& v) {
  const EncLayerP& L = e->den.back();
  FinalArgs f;
  f.P = v.Pf; f.nsplit = v.ffn_slabs; f.pstride = den_slab(e);
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

// shared_qkv: QKV holds ONE sample's projections [T][3D], read by every (sample, head) workgroup (decoder layer 0, dec_layer)
void dec_attention(Ctx& c, int B, int T, const int32_t* lens = nullptr, int shared_qkv = 0) {
  if (!lens) lens = c.e->lens_dev;
  E* e = c.e;
  const int H = e->cfg.num_heads;
  const int nkt = pick_nkt(T);
  dim3 grid(B * H), block(512);
  if (staged_prec(e) != PREC_F32) {
    // the modes that run the decoder GEMMs on bf16 MFMAs run its attention split-bf16 as well (attention.hpp)
    // key-blocked form (40 KB of LDS, two workgroups per CU, any T): pays once there is more than one workgroup per CU to overlap
    // (B H >= 512: 108 vs 133 us at 1 280 workgroups); with one per CU the whole-K/V kernel below is 5 % faster (28.9 vs 30.3 us)
    // (it covers 16 query tiles = 256 frames per (sample, head); longer sequences take the whole-K/V kernel)
    if (T <= 256 && (e->flash_attn == 2 || (e->flash_attn == 1 && B * H >= 512))) {
      // V staged row-major and read as MFMA fragments through ds_read_b64_tr_b16 (r03: 454 -> 417 us per launch at 2 048 motions against
      // transposed V planes written with 2-byte stores; streaming hints on its loads / stores measured level: both alternatives retired in r04)
      MLD_LAUNCH(attn_flash_x3_kernel, grid, block, kFlashLdsBytes, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv);
      count(c);
      check_launch(c, "attn_flash_x3");
      return;
    }
    switch (nkt) {
      case 4: MLD_LAUNCH((attn_decode_x3_kernel<4>), grid, block, attn_x3_lds_bytes<4>(), c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
      case 7: MLD_LAUNCH((attn_decode_x3_kernel<7>), grid, block, attn_x3_lds_bytes<7>(), c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
      case 13: MLD_LAUNCH((attn_decode_x3_kernel<13>), grid, block, attn_x3_lds_bytes<13>(), c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
      default: MLD_LAUNCH((attn_decode_x3_kernel<18>), grid, block, attn_x3_lds_bytes<18>(), c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
    }
    count(c);
    check_launch(c, "attn_decode_x3");
    return;
  }
  const size_t shmem = (size_t)2 * nkt * 16 * 68 * sizeof(float);
  switch (nkt) {
    case 4: MLD_LAUNCH((attn_decode_kernel<4>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
    case 7: MLD_LAUNCH((attn_decode_kernel<7>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
    case 13: MLD_LAUNCH((attn_decode_kernel<13>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
    default: MLD_LAUNCH((attn_decode_kernel<18>), grid, block, shmem, c.stream, (const float*)e->QKV, e->AO, (const int*)lens, T, H, shared_qkv); break;
  }
  count(c);
  check_launch(c, "attn_decode");
}

// Row-strip form of a decoder / encoder GEMM in the split modes (kernels/gemm_strip_x3.hpp) when the shape is one it is built for
// and the weight has a fragment-ordered stream; returns false when the caller should take the staged tiles instead.
// rows per strip of the register-direct decoder kernels: 96 (six row tiles: 2 MB of weights per 96 rows) when the launch fills the chip
// several times over, 64 when it would not -- one bs-64 request is 12 544 frame rows = 131 strips of 96 on 256 CUs, but 196 of 64,
// each a third shorter ("ffn_strip" 1 = this rule, 4 / 6 = always)
int strip_rows_rt(const E* e, int M) {
  if (e->ffn_strip == 4 || e->ffn_strip == 6) return e->ffn_strip;
  return (M + 63) / 64 <= 512 ? 4 : 6;
}

bool strip_gemm(Ctx& c, const GemmArgs& g, bool ln) {
  E* e = c.e;
  const int rt = strip_rows_rt(e, g.M);
  if (!e->strip_gemm || staged_prec(e) != PREC_BF16X3 || e->trace_on || g.M <= e->small_m) return false;
  if (g.K1 != 256 || g.lda != 256 || (g.K2 != 0 && (g.K2 != 256 || g.lda2 != 256)) || g.N % 256 || g.act != ACT_NONE || g.relu_in || g.lens) return false;
  auto it = e->gemm_stream_of.find(g.W);
  if (it == e->gemm_stream_of.end()) return false;
  StripGemmArgs a;
  a.A = g.A; a.A2 = g.A2; a.W = it->second; a.bias = g.bias; a.Y = g.Y; a.ldy = g.ldy; a.M = g.M; a.N = g.N;
  a.skip_lens = g.skip_lens; a.skip_rpg = g.skip_rpg;
  if (ln) {
    if (g.N != 256 || g.K2 != 0 || !g.res || g.ldres != 256 || !g.g1) return false;
    a.res = g.res; a.g1 = g.g1; a.b1 = g.b1; a.cvec = g.cvec; a.rpg = g.rows_per_group; a.g2 = g.g2; a.b2 = g.b2;
    if (g.cvec && (g.ldcvec != 256 || !g.g2)) return false;
    if (rt == 4) MLD_LAUNCH((strip_gemm_x3_kernel<4, 1, true, false>), dim3((g.M + 63) / 64), dim3(512), (strip_gemm_lds_bytes<4, 1, false>()), c.stream, a);
    else MLD_LAUNCH((strip_gemm_x3_kernel<6, 1, true, false>), dim3((g.M + 95) / 96), dim3(512), (strip_gemm_lds_bytes<6, 1, false>()), c.stream, a);
  } else if (g.K2 == 256) {
    if (g.N != 256) return false;
    // (streaming hints measured level on this form -- 414.6 vs 413.7 us, r03c_kernel_stats_ab.csv -- so it has no hinted build)
    MLD_LAUNCH((strip_gemm_x3_kernel<4, 2, false, false>), dim3((g.M + 63) / 64), dim3(512), (strip_gemm_lds_bytes<4, 2, false>()), c.stream, a);
  } else if (rt == 4) {
    // in-projection (N = 768): row strips loaded and outputs stored with the streaming hint (527 -> 504 us per launch at 2 048 motions, r03c)
    MLD_LAUNCH((strip_gemm_x3_kernel<4, 1, false, true, true>), dim3((g.M + 63) / 64), dim3(512), (strip_gemm_lds_bytes<4, 1, true>()), c.stream, a);
  } else {
    MLD_LAUNCH((strip_gemm_x3_kernel<6, 1, false, true, true>), dim3((g.M + 95) / 96), dim3(512), (strip_gemm_lds_bytes<6, 1, true>()), c.stream, a);
  }
  count(c);
  check_launch(c, "strip_gemm_x3");
  return true;
}

// linear1 + GELU + linear2 + residual + LayerNorm of a post-norm layer.  Split-bf16 modes with D = 256, FF = 1024: ONE launch
// (kernels/ffn_strip.hpp) reading its fragment-ordered weight stream; otherwise the two staged GEMMs.  ragged_T > 0: skip all-padding row tiles.
void ffn_block(Ctx& c, const float* x, float* y, int M, const float* w1, const float* b1, const float* w2, const float* b2,
               const float* gamma, const float* beta, int ragged_T) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size;
  if (staged_prec(e) == PREC_BF16X3 && e->ffn_strip && D == 256 && F == 1024 && M > e->small_m && !e->trace_on && e->ffn_stream_of.count(w1)) {
    // register-direct form (kernels/ffn_strip.hpp): weights from the layer's fragment-ordered stream, 96- or 64-row strips
    FfnArgs a;
    a.X = x; a.W1 = e->ffn_stream_of[w1]; a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.Y = y; a.M = M;
    if (ragged_T > 0) { a.skip_lens = e->lens_dev; a.skip_rpg = ragged_T; }
    // auto: 48-row strips, two workgroups per CU (four waves per SIMD, 128 registers each) for launches that fill the chip: 2 % off the
    // decoder against 96-row strips (r03, 2 048 motions: 25.2 vs 25.8 ms) although the weights are streamed twice as often
    if (e->ffn_strip == 3 || (e->ffn_strip == 1 && strip_rows_rt(e, M) == 6)) MLD_LAUNCH(ffn_strip_x3_kernel<3>, dim3((M + 47) / 48), dim3(512), (ffn_strip_lds_bytes<3>()), c.stream, a);
    else if (strip_rows_rt(e, M) == 6) MLD_LAUNCH(ffn_strip_x3_kernel<6>, dim3((M + 95) / 96), dim3(512), (ffn_strip_lds_bytes<6>()), c.stream, a);
    else MLD_LAUNCH(ffn_strip_x3_kernel<4>, dim3((M + 63) / 64), dim3(512), (ffn_strip_lds_bytes<4>()), c.stream, a);
    count(c);
    check_launch(c, "ffn_strip_x3");
    return;
  }
  GemmArgs f1 = lin_args(x, D, D, w1, b1, e->FF, F, M, F);
  f1.act = ACT_GELU;
  GemmArgs f2 = lin_args(e->FF, F, F, w2, b2, y, D, M, D);
  f2.res = x; f2.ldres = D; f2.g1 = gamma; f2.b1 = beta;
  if (ragged_T > 0) {
    f1.skip_lens = f2.skip_lens = e->lens_dev;
    f1.skip_rpg = f2.skip_rpg = ragged_T;
  }
  gemm(c, f1);
  gemm_ln(c, f2);
}

// The decoder's self-attention block on half Q | K | V (kernels/dec_half.hpp; option "dec_half", verdict of finalize's probe in dec_half_ok): split mode,
// row-strip kernels on, D = 256 as 4 heads of 64, at most 16 query tiles per (sample, head)
bool dec_half_on(const E* e, int T) {
  return e->dec_half && (e->dec_half_ok || e->dec_half == 2) && staged_prec(e) == PREC_BF16X3 && e->strip_gemm && !e->trace_on && e->cfg.latent_dim == 256 &&
         e->cfg.num_heads == 4 && T <= 256;
}

// pos_input: xin holds the time queries themselves (zeros + positional rows, init_queries_kernel): row t of EVERY sample is pe[t], so
// the layer's Q, K, V depend on t only.  They are then projected once, for sample 0's T rows, and read by every (sample, head)
// attention workgroup (which still applies its own sample's length mask): exact, and the [B T][3 D] tensor of that layer -- 1.23 GB
// written and read back at 2 048 motions -- never exists ("dec_l0_once").
void dec_layer(Ctx& c, int l, const float* xin, float* xout, int B, int T, bool pos_input = false) {
  E* e = c.e;
  const DecLayerP& L = e->dec[l];
  const int D = e->cfg.latent_dim, M = B * T;
  auto ragged = [&](GemmArgs g) { g.skip_lens = e->lens_dev; g.skip_rpg = T; return g; };   // skip all-padding row tiles
  const bool once = pos_input && e->dec_l0_once && B > 1;
  if (dec_half_on(e, T) && e->gemm_stream_of.count(L.in_w) && (once || M > e->small_m)) {
    // the self-attention block on half Q | K | V (kernels/dec_half.hpp, "dec_half"): in-projection = half rows x split weights, output packed
    // [row][768] halves with q pre-scaled; attention on plain half operands
    unsigned* qh = reinterpret_cast<unsigned*>(e->QKV);
    if (once) {
      // one sample's T rows through the fp32 projection (a launch of a few microseconds), then converted; the halves sit behind the fp32 rows
      // (B > 1: the buffer holds at least two samples' rows)
      gemm(c, lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, T, 3 * D));
      qh += (size_t)T * 3 * D;
      MLD_LAUNCH(qkv_to_half_kernel, dim3((T * 96 + 255) / 256), dim3(256), 0, c.stream, (const float*)e->QKV, qh, T);
      count(c);
      check_launch(c, "qkv_to_half");
    } else {
      InprojHArgs a;
      a.A = xin; a.W = e->gemm_stream_of[L.in_w]; a.bias = L.in_b; a.Y = qh; a.M = M; a.skip_lens = e->lens_dev; a.skip_rpg = T;
      // 64-row strips, two workgroups per CU (69 KB of LDS, 128 registers): one workgroup's row loads / output stores run under the other's products
      // ("dec_half" 6: 96-row strips, one per CU -- a third less weight traffic per row)
      if (e->dec_half == 6) MLD_LAUNCH(strip_inproj_h_kernel<6>, dim3((M + 95) / 96), dim3(512), inproj_h_lds_bytes<6>(), c.stream, a);
      else MLD_LAUNCH(strip_inproj_h_kernel<4>, dim3((M + 63) / 64), dim3(512), inproj_h_lds_bytes<4>(), c.stream, a);
      count(c);
      check_launch(c, "strip_inproj_h");
    }
    MLD_LAUNCH(attn_flash_h_kernel, dim3(B * e->cfg.num_heads), dim3(512), kFlashHLdsBytes, c.stream, (const unsigned*)qh, e->AO, (const int*)e->lens_dev, T, e->cfg.num_heads, once ? 1 : 0);
    count(c);
    check_launch(c, "attn_flash_h");
  } else {
    // once: all T rows (no ragged skip: sample 0 may be shorter than the samples that read its rows)
    const GemmArgs q = once ? lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, T, 3 * D) : ragged(lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, M, 3 * D));
    if (!strip_gemm(c, q, false)) gemm(c, q);
    dec_attention(c, B, T, nullptr, once ? 1 : 0);
  }
  // Chip-filling launches of the split modes: the rest of the layer in ONE launch (kernels/ffn_strip.hpp, TAIL form) -- the H1 tensor
  // between the out-projection kernel and the feed-forward kernel is not written and read back ("dec_tail", on by default)
  if (e->dec_tail && staged_prec(e) == PREC_BF16X3 && e->strip_gemm && (e->ffn_strip == 3 || (e->ffn_strip == 1 && strip_rows_rt(e, M) == 6)) &&
      D == 256 && e->cfg.ff_size == 1024 && !e->trace_on && M > e->small_m && e->ffn_stream_of.count(L.l1_w) && e->gemm_stream_of.count(L.out_w)) {
    FfnArgs a;
    a.W1 = e->ffn_stream_of[L.l1_w]; a.b1 = L.l1_b; a.b2 = L.l2_b; a.gamma = L.n3_w; a.beta = L.n3_b; a.Y = xout; a.M = M;
    a.skip_lens = e->lens_dev; a.skip_rpg = T;
    a.AO = e->AO; a.Wo = e->gemm_stream_of[L.out_w]; a.bo = L.out_b; a.res = xin; a.g1 = L.n1_w; a.be1 = L.n1_b;
    a.cvec = e->cvec + (size_t)l * e->cfg.max_batch * D; a.rpg = T; a.g2 = L.n2_w; a.be2 = L.n2_b;
    // (LDS images row-swizzled like the persistent loop's: 1 476 -> 1 457 us per launch at 2 048 motions, r04a; the plain-image build is retired)
    MLD_LAUNCH((ffn_strip_x3_kernel<3, true, true>), dim3((M + 47) / 48), dim3(512), (ffn_strip_lds_bytes<3>()), c.stream, a);
    count(c);
    check_launch(c, "dec_tail_x3");
    return;
  }
  // out-proj + residual + norm1, then the 1-key cross-attention (a per-sample vector) + norm2
  GemmArgs o = lin_args(e->AO, D, D, L.out_w, L.out_b, e->H1, D, M, D);
  o.res = xin; o.ldres = D; o.g1 = L.n1_w; o.b1 = L.n1_b;
  o.cvec = e->cvec + (size_t)l * e->cfg.max_batch * D; o.ldcvec = D; o.rows_per_group = T;
  o.g2 = L.n2_w; o.b2 = L.n2_b;
  if (!strip_gemm(c, ragged(o), true)) gemm_ln(c, ragged(o));
  ffn_block(c, e->H1, xout, M, L.l1_w, L.l1_b, L.l2_w, L.l2_b, L.n3_w, L.n3_b, T);
}

void skip_linear(Ctx& c, const std::string& prefix, int i, const float* x, const float* skip, float* y, int M, int ragged_T = 0) {
  E* e = c.e;
  const int D = e->cfg.latent_dim;
  GemmArgs g;
  g.A = x; g.lda = D; g.K1 = D; g.A2 = skip; g.lda2 = D; g.K2 = D;
  g.W = P(e, prefix + ".linear_blocks." + std::to_string(i) + ".weight"); g.ldw = 2 * D;
  g.bias = P(e, prefix + ".linear_blocks." + std::to_string(i) + ".bias");
  g.Y = y; g.ldy = D; g.M = M; g.N = D;
  if (ragged_T > 0) { g.skip_lens = e->lens_dev; g.skip_rpg = ragged_T; }   // decoder: skip all-padding row tiles
  if (!strip_gemm(c, g, false)) gemm(c, g);
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
      dec_layer(c, l, xin, xout, B, T, l == 0);
      xin = xout;
    }
    GemmArgs f = lin_args(xin, D, D, P(e, "vae.decoder.final_layer.weight"), P(e, "vae.decoder.final_layer.bias"), feats_out, NF, M, NF);
    f.lens = e->lens_dev; f.rows_per_group = T;   // output[~mask.T] = 0 (actor_vae.py:231)
    gemm(c, f);
    return;
  }
  const float* x = e->X0;
  for (int l = 0; l < nb; ++l) {
    dec_layer(c, l, x, e->S[l], B, T, l == 0);
    x = e->S[l];
  }
  dec_layer(c, nb, x, e->Ha, B, T, nb == 0);
  for (int i = 0; i < nb; ++i) {
    skip_linear(c, "vae.decoder", i, e->Ha, e->S[nb - 1 - i], e->Hb, M, T);
    dec_layer(c, nb + 1 + i, e->Hb, e->Ha, B, T);
  }
  if (e->final_stream && staged_prec(e) == PREC_BF16X3 && D == 256 && !e->trace_on && M > e->small_m) {
    // decoder.norm + final_layer + output[~mask.T] = 0 as one row-strip launch (kernels/final_strip.hpp, "final_strip")
    FinalStripArgs a;
    a.X = e->Ha; a.gamma = P(e, "vae.decoder.norm.weight"); a.beta = P(e, "vae.decoder.norm.bias"); a.W = e->final_stream;
    a.bias = P(e, "vae.final_layer.bias"); a.Y = feats_out; a.M = M; a.NF = NF; a.lens = e->lens_dev; a.rpg = T;
    MLD_LAUNCH(final_strip_x3_kernel, dim3((M + kFinalStripRows - 1) / kFinalStripRows), dim3(512), final_strip_lds_bytes(), c.stream, a);
    count(c);
    check_launch(c, "final_strip_x3");
    return;
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
  const int D = e->cfg.latent_dim, M = B * S;
  {
    const GemmArgs q = lin_args(xin, D, D, L.in_w, L.in_b, e->QKV, 3 * D, M, 3 * D);
    if (!strip_gemm(c, q, false)) gemm(c, q);
  }
  dec_attention(c, B, S, e->lens2_dev);
  GemmArgs o = lin_args(e->AO, D, D, L.out_w, L.out_b, e->H1, D, M, D);
  o.res = xin; o.ldres = D; o.g1 = L.n1_w; o.b1 = L.n1_b;
  if (!strip_gemm(c, o, true)) gemm_ln(c, o);
  ffn_block(c, e->H1, xout, M, L.l1_w, L.l1_b, L.l2_w, L.l2_b, L.n2_w, L.n2_b, 0);
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

// Everything mld.py:232-240,264 does after the text encoder (enqueue_sample below).  The whole CFG batch runs as ONE
// chain of dependent launches (splitting a batch into sub-batch chains on parallel graph branches was measured: the
// sequential depth per chain is what costs, no gain -- profiles/r01_v3_chains*; removed).
// rows of token 2 for an action CFG batch of R rows -> dst[R][D] (labels already in labels_dev)
void action_rows(Ctx& c, int R, int nuncond, float* dst) {
  E* e = c.e;
  MLD_LAUNCH(action_rows_kernel, dim3(R), dim3(256), 0, c.stream, dst, P(e, "denoiser.emb_proj.action_embedding"),
             P(e, "denoiser.query_pos.pe") + 2 * e->cfg.latent_dim, (const int*)e->labels_dev, nuncond);
  count(c);
  check_launch(c, "action_rows");
}

// run-time part of the F16X3 range contract (mldhip.h): non-finite results are counted, mldhip_numeric_status reports them
void count_nonfinite(Ctx& c, const float* x, long long n) {
  E* e = c.e;
  if (e->cfg.precision != MLDHIP_PREC_BF16X3_DECODE || !e->nonfinite || n <= 0) return;
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 2048);
  MLD_LAUNCH(count_nonfinite_kernel, dim3(blocks), dim3(256), 0, c.stream, x, n, e->nonfinite);
  check_launch(c, "count_nonfinite");
}

// the whole reverse loop (or its first `n` steps: finalize's range probe) as one persistent launch: a workgroup per 8 motions (kernels/loop_fused.hpp)
void launch_fused_loop(Ctx& c, const float* init_lat, int B, int n, float guidance) {
  E* e = c.e;
  LoopArgs a;
  const bool x3 = fused_split(e);
  a.stream = x3 ? e->loop_stream_x3 : e->loop_stream; a.ips = e->loop_ips; a.small = e->loop_small; a.T1 = e->T1; a.TP = e->TP; a.init_lat = init_lat;
  a.lat = e->lat; a.skip = e->FS; a.ddim = e->loop_ddim; a.B = B; a.L = e->cfg.num_layers; a.n = n;
  a.guidance = guidance; a.init_sigma = 1.0f;
  const dim3 grid((B + 7) / 8);
#if defined(MLDHIP_HOOKS)
  if (x3 && e->fused_dbg == 5) { a.trace = reinterpret_cast<unsigned long long*>(e->trace_buf); MLD_LAUNCH((den_loop_kernel<true, 5>), grid, dim3(512), kLoopLdsBytes, c.stream, a); }
  else
#endif
  if (x3) MLD_LAUNCH((den_loop_kernel<true>), grid, dim3(512), kLoopLdsBytes, c.stream, a);
  else MLD_LAUNCH((den_loop_kernel<false>), grid, dim3(512), kLoopLdsBytes, c.stream, a);
  count(c);
  check_launch(c, "den_loop");
}

// the whole reverse loop (or its first `n` steps) of up to 8 x kClMaxClusters motions as one launch of clusters (kernels/loop_cluster.hpp)
// motions [s_base, s_base + nm) of a call of B
void launch_cluster_chunk(Ctx& c, const float* init_lat, int B, int s_base, int nm, int n, float guidance) {
  E* e = c.e;
  ClusterArgs a;
  const int cg = cluster_groups(e, nm), members = 3 * cg;
  a.s_base = s_base; a.s_end = s_base + nm;
  a.timeout = e->cluster_timeout ? (unsigned)e->cluster_timeout : kClTimeoutTicks; a.mute = e->cluster_mute;
  a.stream = e->cl_stream;
  a.wave_off = e->cl_wave_off_dev + (cg == 8 ? 32 : 0);
  a.small = e->loop_small; a.T1 = e->T1; a.TP = e->TP; a.init_lat = init_lat; a.lat = e->lat; a.park = e->cl_park; a.ddim = e->loop_ddim;
  a.xbuf = e->cl_xbuf;
  a.host_status = e->cl_host_status;
  a.ncl = (nm + 7) / 8;
  a.flags = reinterpret_cast<unsigned*>(e->cl_flags);
  a.status = a.flags + (size_t)std::min<size_t>(kClMaxClusters, (e->cfg.max_batch + 7) / 8) * kClFlagWords;
  a.B = B; a.L = e->cfg.num_layers; a.n = n; a.guidance = guidance; a.init_sigma = 1.0f;
#if defined(MLDHIP_SIM)
  a.xslots = std::min(a.ncl, 8);          // the simulator creates a fiber per work-item of every block: no idle XCD slots
#else
  a.xslots = 8;                           // block b -> XCD b % 8 (observed placement): a cluster's members share a slot
#endif
  // every polled word is zero at the start of every call (Guideline 16 "Re-initialise every call")
  const int words = (int)(a.status - a.flags) + 2;        // the flags and the two per-launch status words (status[2] is sticky: cluster_timed_out)
  if (e->sample_part != 2)
  MLD_LAUNCH(clear_cluster_flags_kernel, dim3(1), dim3(256), 0, c.stream, a.flags, words);      // (a kernel, NOT a memset node: replays of a captured hipMemsetAsync left address-like words here on this runtime, DESIGN.md 3a -- the entry check of the kernel now catches such a launch)
#if defined(MLDHIP_HOOKS)
  if (e->cluster_stale && e->sample_part != 2) MLD_LAUNCH(poke_cluster_flag_kernel, dim3(1), dim3(1), 0, c.stream, a.flags + kFlagH * kClFlagLine + 3, 77u);
#endif
  if (e->sample_part == 1) return;               // (the pipelined form captures what precedes the launch as a graph of its own)
  const dim3 grid((unsigned)(a.xslots * members * ((a.ncl + a.xslots - 1) / a.xslots)));
  if (cg == 8) {
    if (e->cluster_wt) MLD_LAUNCH_CORESIDENT((den_cluster_kernel<true, 8>), grid, dim3(512), kClLdsBytes, c.stream, a);
    else MLD_LAUNCH_CORESIDENT((den_cluster_kernel<false, 8>), grid, dim3(512), kClLdsBytes, c.stream, a);
  } else {
    if (e->cluster_wt) MLD_LAUNCH_CORESIDENT((den_cluster_kernel<true, 4>), grid, dim3(512), kClLdsBytes, c.stream, a);
    else MLD_LAUNCH_CORESIDENT((den_cluster_kernel<false, 4>), grid, dim3(512), kClLdsBytes, c.stream, a);
  }
  count(c);
  check_launch(c, "den_cluster");
}

// up to 128 motions: one launch; up to kClMaxCall = 256: two launches one after the other on the call's stream (they share the exchange regions and flags; 2 x 7.6 ms
// against the sample-major loop's flat 18.7 ms) -- never side by side: 2 x 192 workgroups are not co-resident
void launch_cluster_loop(Ctx& c, const float* init_lat, int B, int n, float guidance) {
  const int chunk = c.e->cluster_chunk;          // 128 (hooks / simulator builds: "cluster_chunk" makes the two-launch path testable on a few motions)
  for (int s = 0; s < B && !c.rc; s += chunk) launch_cluster_chunk(c, init_lat, B, s, std::min(chunk, B - s), n, guidance);
}

// the decode half of a sample call: MldVae.decode of the bound context's latents + feats2joints (mld.py:232-240,264), with the run-time non-finite count
void enqueue_decode(Ctx& c, int B, int T, float* feats_out, float* joints_out) {
  E* e = c.e;
  e->phase = 1;
  float* f = feats_out ? feats_out : e->feats_int;
  decode_body(c, e->lat, B, T, f);
  if (joints_out) {
    e->phase = 2;
    joints_body(c, f, B, T, joints_out);
    count_nonfinite(c, joints_out, (long long)B * T * e->cfg.njoints * 3);
  } else {
    count_nonfinite(c, f, (long long)B * T * e->cfg.nfeats);      // feats-only call: the decoder's output is what the caller gets
  }
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
  if (e->sample_part != 2) {
    if (text) text_projection(c, text, 2 * B, e->TP);
    else action_rows(c, 2 * B, B, e->TP);
  }
  if (use_cluster(e, B)) {
    launch_cluster_loop(c, init_lat, B, n, guidance);
    if (e->sample_part == 1) return c.rc;
  } else if (use_fused(e, B)) {
    launch_fused_loop(c, init_lat, B, n, guidance);
  } else {
    const DenView v = den_view(e, 2 * B);
    MLD_LAUNCH(init_chain_kernel, dim3(B), dim3(256), 0, stream, init_lat, v.lat, v.X0, P(e, "denoiser.query_pos.pe"),
               (const float*)e->T1, (const float*)e->TP, B, 0, B, 1.0f /* init_noise_sigma */);
    count(c);
    check_launch(c, "init_chain");
    for (int s = 0; s < n && !c.rc; ++s) {
      denoiser_body(c, v);
      const float* t1n = (s + 1 < n) ? e->T1 + (size_t)(s + 1) * D : nullptr;
      MLD_LAUNCH(den_final_step_kernel, dim3(B), dim3(256), 0, stream, den_final_args(e, v), v.lat, v.X0,
                 P(e, "denoiser.query_pos.pe"), t1n, B, guidance, ddim_coef(e, e->timesteps[s]));
      count(c);
      check_launch(c, "den_final_step");
    }
  }
  if (c.rc || e->sample_part == 2) return c.rc;    // (part 2: the launch alone; the pipelined form counts non-finite latents on its side stream, in front of the decode)
  count_nonfinite(c, e->lat, (long long)B * D);
  if (lat_out) {
    hipError_t s = hipMemcpyAsync(lat_out, e->lat, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream);
    if (s != hipSuccess) return e->fail(MLDHIP_EHIP, "latents copy: %s", hipGetErrorString(s));
  }
  if (feats_out || joints_out) enqueue_decode(c, B, T, feats_out, joints_out);
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
