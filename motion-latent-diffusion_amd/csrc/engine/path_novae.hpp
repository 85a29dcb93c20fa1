// Diffusion-only path (config 4): trans_dec denoiser on raw motion + DDPM.
// Part of libmldhip's single translation unit (included by ../mldhip.hip, in this order: state, params, dispatch,
// path_latent, path_novae).  Internal linkage throughout (anonymous namespace) except the handle type itself.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Diffusion-only variant (BASELINE config 4): trans_dec denoiser on raw motion, d = 512 (kernels/novae.hpp).
// Row layout: sample-major rows r*T + t of the CFG batch (r < R = 2B), 512 floats per row.
void novae_ln(Ctx& c, const float* x, const float* res, const float* g, const float* b, float* y, int M) {
  MLD_LAUNCH((add_layernorm_rows_kernel<512>), dim3((M + 3) / 4), dim3(256), 0, c.stream, x, res, g, b, y, M);
  count(c);
  check_launch(c, "add_layernorm_rows");
}

void novae_self_attention(Ctx& c, int R, int T, bool force_flash = false) {
  E* e = c.e;
  const int H = e->cfg.num_heads, nkt = pick_nkt(T), nqt = (T + 15) / 16;
  dim3 grid(R * H, (nqt + 7) / 8), block(512);
  const int* nolens = nullptr;    // the reference passes no key-padding mask to the trans_dec denoiser (mld_denoiser.py:215)
  if (staged_prec(e) == PREC_BF16X3 && T <= 256 && (e->flash_attn == 2 || (e->flash_attn == 1 && (R * H >= 512 || force_flash))) && e->cfg.latent_dim == H * 128) {
    // key-blocked form (attention.hpp attn_flash128_x3_kernel): one workgroup per (sample, head), K / V in blocks of 32 keys
    MLD_LAUNCH(attn_flash128_x3_kernel, dim3(R * H), dim3(512), kFlash128LdsBytes, c.stream, (const float*)e->QKV, e->AO, nolens, T, H);
    count(c);
    check_launch(c, "attn_flash128_x3");
    return;
  }
  if (staged_prec(e) != PREC_F32) {   // GEMMs on bf16 MFMAs: the attention runs split-bf16 too
    switch (nkt) {
      case 4: MLD_LAUNCH((attn_seq_x3_kernel<4, 128>), grid, block, (attn_seq_x3_lds_bytes<4, 128>()), c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
      case 7: MLD_LAUNCH((attn_seq_x3_kernel<7, 128>), grid, block, (attn_seq_x3_lds_bytes<7, 128>()), c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
      case 13: MLD_LAUNCH((attn_seq_x3_kernel<13, 128>), grid, block, (attn_seq_x3_lds_bytes<13, 128>()), c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
      default: MLD_LAUNCH((attn_seq_x3_kernel<18, 128>), grid, block, (attn_seq_x3_lds_bytes<18, 128>()), c.stream, (const float*)e->QKV, e->AO, nolens, T, H); break;
    }
    count(c);
    check_launch(c, "attn_seq_x3");
    return;
  }
  const size_t shmem = (size_t)nkt * 16 * 132 * sizeof(float);
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

// Memory tokens folded through the cross-attention's query and out projections (kernels/novae.hpp cross_fold_kernel): kv [L][ntok][2D] -> w, u [L][ntok][H][D], c [L][ntok][H]
void novae_fold_memory(Ctx& c, const float* kv, int ntok, long long kv_layer_stride, float* w, float* u, float* cc) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, H = e->cfg.num_heads, L = e->cfg.num_layers;
  if (D != 512 || H != 4) return;                  // (the folded form is built for config 4's shape; cross_fold_on() says the same)
  MLD_LAUNCH((cross_fold_kernel<512, 128>), dim3((unsigned)ntok, (unsigned)H, (unsigned)L), dim3(256), kCrossFoldLdsBytes, c.stream, kv, kv_layer_stride,
             e->ndec[0].cin_w, e->ndec[0].cin_b, e->ndec[0].cout_w, (long long)e->ndec_layer_stride, w, u, (long long)ntok * H * D, cc, (long long)ntok * H, ntok);
  count(c);
  check_launch(c, "cross_fold");
}
bool cross_fold_on(const E* e) { return e->cross_fold && e->cfg.latent_dim == 512 && e->cfg.num_heads == 4 && !e->trace_on; }

// the time token's folded vectors of this call: layer l at w + l * tokens * H * D (u likewise), c + l * tokens * H
struct NovaeFold { const float* w; const float* u; const float* c; long long tokens; };

// MldDenoiser.forward, trans_dec branch, for the M = R*T rows whose zero-padded features are in e->FF [M][KP].
// tkv: K|V of the time token, layer l at tkv + l*tkv_stride; text-token K|V in e->XKV [L][2*max_batch][2D].
// cfg_dup: the rows of samples [R / 2, R) are copies of those of [0, R / 2) (torch.cat([latents] * 2), mld.py:325) and only R / 2 samples' features are in e->FF -- the two CFG
// halves differ in their TEXT token only, which enters at the first cross-attention: the embedding, layer 0's in-projection, self-attention and out-projection run on half the
// rows and the folded cross-attention launch of layer 0 reads them for both halves (needs "cross_fold"; novae_cfg_dedup() decides)
bool novae_cfg_dedup(const E* e, int R) { return cross_fold_on(e) && R % 2 == 0 && R >= 2; }
void novae_denoiser_body(Ctx& c, int R, int T, const float* tkv, long long tkv_stride, const NovaeFold& tf, float* eps_out, bool cfg_dup = false) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, F = e->cfg.ff_size, NF = e->cfg.nfeats, KP = novae_kp(e), M = R * T;
  const bool dup = cfg_dup && novae_cfg_dedup(e, R);
  const int M0 = dup ? M / 2 : M, R0 = dup ? R / 2 : R;              // rows / samples in front of layer 0's cross-attention
  gemm(c, lin_args(e->FF, KP, KP, e->WskelP, P(e, "denoiser.pose_embd.bias"), e->X0, D, M0, D));
  MLD_LAUNCH(add_pe_mod_kernel, dim3(std::min(4096, (M0 * (D / 4) + 255) / 256)), dim3(256), 0, c.stream, e->X0,
             P(e, "denoiser.query_pos.pe"), (long long)M0, T, D);
  count(c);
  check_launch(c, "add_pe_mod");
  for (int l = 0; l < e->cfg.num_layers && !c.rc; ++l) {
    const DecLayerP& L = e->ndec[l];
    const bool half = dup && l == 0;
    const int Ml = half ? M0 : M;
    float* xc = e->X0;                                                 // the layer's stream after the cross-attention block (layer 0 of a de-duplicated batch: H1, the launch must not write where the other half still reads)
    gemm(c, lin_args(e->X0, D, D, L.in_w, L.in_b, e->QKV, 3 * D, Ml, 3 * D));
    novae_self_attention(c, half ? R0 : R, T, half);
    gemm(c, lin_args(e->AO, D, D, L.out_w, L.out_b, e->Ha, D, Ml, D));
    if (cross_fold_on(e)) {
      // LayerNorm 1 + the whole two-token cross-attention sub-layer + LayerNorm 2 in one launch, on the folded memory tokens (kernels/novae.hpp): no query GEMM, no
      // out-projection GEMM, two LayerNorm passes less -- 5 launches and 26 GFLOP of a layer's 143 become ~30 us of streaming
      const size_t H = (size_t)e->cfg.num_heads, Bm2 = (size_t)2 * e->cfg.max_batch;
      Cross2LnArgs a;
      a.Ha = e->Ha; a.X = e->X0; a.g1 = L.n1_w; a.b1 = L.n1_b;
      a.wt = tf.w + (size_t)l * tf.tokens * H * D; a.ut = tf.u + (size_t)l * tf.tokens * H * D; a.ct = tf.c + (size_t)l * tf.tokens * H;
      a.wx = e->XKW + (size_t)l * Bm2 * H * D; a.ux = e->XKU + (size_t)l * Bm2 * H * D; a.cx = e->XKC + (size_t)l * Bm2 * H;
      if (half) { xc = e->H1; a.src_mod = R0; }
      a.bo = L.cout_b; a.g2 = L.n2_w; a.b2 = L.n2_b; a.Y = xc; a.M = M; a.T = T;
      MLD_LAUNCH((cross2_fold_ln_kernel<512, 128>), dim3((unsigned)(R * ((T + kC2Rows - 1) / kC2Rows))), dim3(256), kC2LdsBytes, c.stream, a);
      count(c);
      check_launch(c, "cross2_fold_ln");
    } else {
    novae_ln(c, e->Ha, e->X0, L.n1_w, L.n1_b, e->H1, M);
    gemm(c, lin_args(e->H1, D, D, L.cin_w, L.cin_b, e->Hb, D, M, D));                    // cross-attention queries
    MLD_LAUNCH((cross2_kernel<512, 128>), dim3((M + 3) / 4), dim3(256), 0, c.stream, (const float*)e->Hb, tkv + (size_t)l * tkv_stride,
               (const float*)(e->XKV + (size_t)l * 2 * e->cfg.max_batch * 2 * D), e->AO, M, T);
    count(c);
    check_launch(c, "cross2");
    gemm(c, lin_args(e->AO, D, D, L.cout_w, L.cout_b, e->Ha, D, M, D));
    novae_ln(c, e->Ha, e->H1, L.n2_w, L.n2_b, e->X0, M);
    }
    GemmArgs f1 = lin_args(xc, D, D, L.l1_w, L.l1_b, e->FF, F, M, F);
    f1.act = ACT_GELU;
    gemm(c, f1);
    gemm(c, lin_args(e->FF, F, F, L.l2_w, L.l2_b, e->Ha, D, M, D));
    novae_ln(c, e->Ha, xc, L.n3_w, L.n3_b, e->X0, M);          // (in place when xc == X0: a wave reads its whole row before writing it)
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

void novae_fold_text(Ctx& c, int rows) {
  E* e = c.e;
  const int D = e->cfg.latent_dim, H = e->cfg.num_heads, L = e->cfg.num_layers;
  const long long tok = (long long)2 * e->cfg.max_batch;
  MLD_LAUNCH((cross_fold_kernel<512, 128>), dim3((unsigned)rows, (unsigned)H, (unsigned)L), dim3(256), kCrossFoldLdsBytes, c.stream, (const float*)e->XKV, tok * 2 * D,
             e->ndec[0].cin_w, e->ndec[0].cin_b, e->ndec[0].cout_w, (long long)e->ndec_layer_stride, e->XKW, e->XKU, tok * H * D, e->XKC, tok * H, rows);
  count(c);
  check_launch(c, "cross_fold_text");
}

// text token of the memory: emb_proj(text) + mem_pos.pe[1] -> TP [rows][D], then its K|V for every layer -> XKV
void novae_text_memory(Ctx& c, const float* text, int rows) {
  E* e = c.e;
  text_projection(c, text, rows, e->TP);
  novae_memory_kv(c, e->TP, rows, e->XKV, (long long)2 * e->cfg.max_batch * 2 * e->cfg.latent_dim);
  // ... and their folded forms: ONCE per call, the text tokens do not depend on the step.  (Token r of layer l sits at [l][r] of buffers laid out for 2 * max_batch tokens.)
  if (cross_fold_on(e)) novae_fold_text(c, rows);
}

// MLD.forward after the text encoder with vae_type 'no' (mld.py:232-242,264,290-360), in three pieces so that the step
// loop can be captured in graph chunks.  lens_dev holds lengths ++ lengths.
int novae_prologue(E* e, hipStream_t stream, const float* text, const float* init_lat, int B, int T) {
  Ctx c{e, stream};
  const long long nel = (long long)B * T * e->cfg.nfeats;
  e->launches[0] = e->launches[1] = e->launches[2] = 0;
  e->phase = 0;
  HIP_TRY(e, hipMemcpyAsync(e->lat, init_lat, nel * sizeof(float), hipMemcpyDeviceToDevice, stream));   // init_noise_sigma = 1
  novae_text_memory(c, text, 2 * B);
  return c.rc;
}

// DDPM steps [s0, s1): CFG batch through the trans_dec denoiser, guidance, ancestral step.  The per-step Gaussian draw is
// step_noise[s] when injected, else Philox(seed, s, element) with the seed taken from *seed_dev when that is non-null.
int novae_steps(E* e, hipStream_t stream, int B, int T, int s0, int s1, const float* step_noise, unsigned long long seed,
                const unsigned long long* seed_dev) {
  Ctx c{e, stream};
  const int D = e->cfg.latent_dim, NF = e->cfg.nfeats, n = e->cfg.num_inference_steps;
  const long long nel = (long long)B * T * NF;
  const float guidance = e->cfg.guidance_scale > 1.0f ? e->cfg.guidance_scale : 1.0f;
  e->phase = 0;
  for (int s = s0; s < s1 && !c.rc; ++s) {
    const bool dedup = novae_cfg_dedup(e, 2 * B);
    novae_pad_input(c, e->lat, (long long)B * T, dedup ? 1 : 2);                          // torch.cat([latents] * 2) (one copy when the halves are de-duplicated)
    const size_t Hn = (size_t)e->cfg.num_heads;
    novae_denoiser_body(c, 2 * B, T, e->TKV + (size_t)s * 2 * D, (long long)n * 2 * D, NovaeFold{e->TKW + (size_t)s * Hn * D, e->TKU + (size_t)s * Hn * D, e->TKC + (size_t)s * Hn, n}, e->feats_int, dedup);
    MLD_LAUNCH(cfg_ddpm_step_kernel, dim3((unsigned)std::min<long long>(4096, (nel / 4 + 255) / 256)), dim3(256), 0, stream,
               (const float*)e->feats_int, (const float*)(e->feats_int + nel), (const float*)e->lat,
               step_noise ? step_noise + (size_t)s * nel : (const float*)nullptr, e->lat, nel, guidance,
               ddpm_coef(e, e->timesteps[s]), seed, (unsigned)s, seed_dev);
    count(c);
    check_launch(c, "cfg_ddpm_step");
  }
  return c.rc;
}

int novae_epilogue(E* e, hipStream_t stream, int B, int T, float* feats_out, float* joints_out) {
  Ctx c{e, stream};
  const long long nel = (long long)B * T * e->cfg.nfeats;
  count_nonfinite(c, e->lat, nel);       // the diffusion-only variant's result IS the sampled motion (advisor r4: it was never counted)
  if (feats_out) HIP_TRY(e, hipMemcpyAsync(feats_out, e->lat, nel * sizeof(float), hipMemcpyDeviceToDevice, stream));   // "decode" = identity (mld.py:241-242)
  if (joints_out) {
    e->phase = 2;
    joints_body(c, e->lat, B, T, joints_out);
    count_nonfinite(c, joints_out, (long long)B * T * e->cfg.njoints * 3);
  }
  return c.rc;
}

}  // namespace
