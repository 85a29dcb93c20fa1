// Kernels of the no-VAE ("diffusion only") variant: raw-motion diffusion with the trans_dec denoiser
// (configs/config_novae_humanml3d.yaml; mld_denoiser.py:208-221 -> TransformerDecoder, cross_attention.py:195-233,
// layers :323-345) and the DDPM scheduler (configs/modules_novae/scheduler.yaml:16-29).
//
// Shapes (BASELINE config 4): d = 512, 4 heads x 128, ff 1024, 9 layers, R = 2B CFG rows x T frames: M = 25 088 rows.
// 1.29 TFLOP per step, x1000 steps: the GEMMs (gemm.hpp, staged exact-fp32 MFMA) carry > 90 % of the time, so the
// row-wise pieces here are plain HBM passes (each moves 100-150 MB per call, ~1 % of a step) -- not fused into the GEMM
// epilogues on purpose in this first version.
#pragma once
#include "rt.hpp"

namespace mld {

// out[(h*B + b)*T + t][0:KP] = lat[b][t][0:NF] padded with zeros, h = 0,1 (torch.cat([latents] * 2), mld.py:324-326);
// `dup` = 1 copies once (per-op entry point: the caller already passes R rows).
__global__ __launch_bounds__(256) void dup_pad_rows_kernel(const float* __restrict__ lat, float* __restrict__ out,
                                                           long long rows, int NF, int KP, int dup) {
  const long long n = rows * KP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / KP;
    const int c = int(i - row * KP);
    const float v = c < NF ? lat[row * NF + c] : 0.f;
    for (int h = 0; h < dup; ++h) out[(h * rows + row) * KP + c] = v;
  }
}

// H[row][:] += pe[row % T][:]   (PositionEmbeddingLearned1D over the frame axis, position_encoding.py:153-159)
__global__ __launch_bounds__(256) void add_pe_mod_kernel(float* __restrict__ H, const float* __restrict__ pe, long long M, int T,
                                                         int D) {
  const long long n4 = M * D / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4, row = e / D;
    const int d = int(e - row * D), t = int(row % T);
    F4 a = ld4(H + e);
    const F4 p = ld4(pe + (long long)t * D + d);
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    st4(H + e, a);
  }
}

// Y = LayerNorm(X + res) * gamma + beta over rows of width W (res may be null).  One wave per row, 4 rows per
// workgroup; a lane owns W/256 float4 chunks (chunk c at column c*256 + lane*4: every wave load is one 1 KiB line).
template <int W>
__global__ __launch_bounds__(256) void add_layernorm_rows_kernel(const float* __restrict__ X, const float* __restrict__ res,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ Y, int M) {
  constexpr int NC = W / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row = blockIdx.x * 4 + wave;
  const bool live = row < M;
  row = live ? row : M - 1;
  F4 x[NC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) x[c] = ld4(X + (long long)row * W + c * 256 + lane * 4);
  if (res) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const F4 r = ld4(res + (long long)row * W + c * 256 + lane * 4);
      x[c].x += r.x; x[c].y += r.y; x[c].z += r.z; x[c].w += r.w;
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
  const float mean = sum64(s) * (1.0f / W);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    x[c].x -= mean; x[c].y -= mean; x[c].z -= mean; x[c].w -= mean;
    q += (x[c].x * x[c].x + x[c].y * x[c].y) + (x[c].z * x[c].z + x[c].w * x[c].w);
  }
  const float rstd = rsqrtf(sum64(q) * (1.0f / W) + kLnEps);
  if (!live) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const F4 gm = ld4(gamma + c * 256 + lane * 4), bt = ld4(beta + c * 256 + lane * 4);
    F4 y;
    y.x = x[c].x * rstd * gm.x + bt.x; y.y = x[c].y * rstd * gm.y + bt.y;
    y.z = x[c].z * rstd * gm.z + bt.z; y.w = x[c].w * rstd * gm.w + bt.w;
    st4(Y + (long long)row * W + c * 256 + lane * 4, y);
  }
}

// ----------------------------------------------------------------------------------------------
// Self-attention over the T frames of one (sample, head) for head dims that do not leave room for K AND V in LDS
// (HD = 128: 208 keys x 132 floats = 107 KiB per operand).  Same MFMA mapping as attn_decode_kernel (swapped QK^T,
// P registers as the A operand of P.V) but K and V pass through ONE LDS buffer in turn, and a workgroup owns only
// NW query tiles (one per wave) so the scores of its queries stay in registers across the K -> V switch.
//   grid = (samples * H, ceil(ceil(T/16) / NW)), block = 64*NW, dynamic LDS = NKT*16*(HD+4)*4 bytes.
// NW = 8: two waves per SIMD (one wave's softmax / LDS reads overlap the other's MFMAs) and K/V staged half as often.
// lens == nullptr: no key-padding mask (the trans_dec denoiser attends to all T frames, mld_denoiser.py:215).
template <int NKT, int HD, int NW = 8>
__global__ __launch_bounds__(NW * 64) void attn_seq_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                       const int* __restrict__ lens, int T, int H) {
  constexpr int ST = HD + 4, C4 = HD / 4, KPL = HD / 4;   // LDS row stride; float4 chunks per row; q/k dims per lane
#if defined(MLDHIP_SIM)
  float* KV = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* KV = smem;
#endif
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  int len = T;
  if (lens) len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nqt = (T + 15) >> 4;
  const int qt = blockIdx.y * NW + wave;
  const bool active = qt < nqt;                       // idle waves still take part in the barriers
  const float scale = rsqrtf((float)HD);

  // K (col0 = D) or V (col0 = 2D) rows of this (sample, head) -> LDS.  Every thread issues ALL its global loads of a
  // half-pass before the first LDS store (a load-store-per-iteration loop pays the full memory latency per iteration:
  // 26 x ~1 us per operand, 3x the MFMA time of the whole workgroup -- profiles/r01_v12_novae_kernel_stats.csv).
  // Addresses are clamped and rows >= len zeroed by a multiply (P is 0 there and 0*garbage must not be NaN), so no
  // load sits behind a branch.
  auto stage = [&](int col0) {
    constexpr int KPI = NW * 64 / C4, NIT = (NKT * 16 + KPI - 1) / KPI, UB = (NIT + 1) / 2;
    const int c4 = tid % C4, k0 = tid / C4;
    const float* base = qkv + (long long)b * T * 3 * D + col0 + h * HD + c4 * 4;
#pragma unroll
    for (int j0 = 0; j0 < NIT; j0 += UB) {
      F4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int key = (j0 + u) * KPI + k0;
        const int kc = key < len ? key : len - 1;
        v[u] = ld4(base + (long long)kc * 3 * D);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int key = (j0 + u) * KPI + k0;
        const float m = key < len ? 1.f : 0.f;
        if (j0 + u < NIT && key < nkt * 16) st4(KV + key * ST + c4 * 4, F4{v[u].x * m, v[u].y * m, v[u].z * m, v[u].w * m});
      }
    }
  };

  stage(D);
  __syncthreads();
  f32x4 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float inv = 0.f;
  if (active) {
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = qkv + (long long)(b * T + qrow) * 3 * D + h * HD + g * KPL;
    float qf[KPL];
#pragma unroll
    for (int c = 0; c < KPL / 4; ++c) {
      const F4 t = ld4(qp + c * 4);
      qf[c * 4] = t.x * scale; qf[c * 4 + 1] = t.y * scale; qf[c * 4 + 2] = t.z * scale; qf[c * 4 + 3] = t.w * scale;
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        const float* kp = KV + (kt * 16 + r) * ST + g * KPL;
#pragma unroll
        for (int c = 0; c < KPL / 4; ++c) {
          const F4 t = ld4(kp + c * 4);
          s[kt] = mfma_f32_16x16x4(t.x, qf[c * 4], s[kt]);
          s[kt] = mfma_f32_16x16x4(t.y, qf[c * 4 + 1], s[kt]);
          s[kt] = mfma_f32_16x16x4(t.z, qf[c * 4 + 2], s[kt]);
          s[kt] = mfma_f32_16x16x4(t.w, qf[c * 4 + 3], s[kt]);
        }
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (kt < nkt) && (kt * 16 + g * 4 + i < len);
        s[kt][i] = valid ? s[kt][i] : -INFINITY;
        m = fmaxf(m, s[kt][i]);
      }
    m = max_groups(m);
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = expf(s[kt][i] - m);
        s[kt][i] = e;
        den += e;
      }
    den = sum_groups(den);
    inv = 1.0f / den;
  }
  __syncthreads();                                    // every wave is done reading K
  stage(2 * D);
  __syncthreads();
  if (!active) return;
  f32x4 oacc[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pv = s[kt][i] * inv;
        const float* vp = KV + (kt * 16 + g * 4 + i) * ST + r;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) oacc[dt] = mfma_f32_16x16x4(pv, vp[dt * 16], oacc[dt]);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = qt * 16 + g * 4 + i;
      if (q < T) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[dt][i];
    }
}

// ----------------------------------------------------------------------------------------------
// Split-bf16 form of attn_seq_kernel (the arithmetic modes that run this variant's GEMMs on bf16 MFMAs): same two-phase
// structure -- K, then V, through ONE LDS buffer, a workgroup's NW query tiles keep their scores in registers across the
// switch -- with the operand layouts of attn_decode_x3_kernel (attention.hpp): K as two bf16 planes [key][HD], V transposed
// as two bf16 planes [dim][keys] so that a lane's eight k-slots of a 32-key block are the keys it already holds scores for.
// 3 bf16 MFMAs (lo*hi, hi*lo, hi*hi) per 32-wide contraction chunk, fp32 accumulate; softmax and scaling in fp32.
template <int NKT, int HD>
constexpr int attn_seq_x3_lds_bytes() {
  constexpr int kwords = 2 * NKT * 16 * (HD / 2 + (NKT <= 13 ? 8 : 4)), vwords = 2 * HD * (((NKT + 1) / 2) * 16 + 4);
  return (kwords > vwords ? kwords : vwords) * 4;
}

template <int NKT, int HD, int NW = 8>
__global__ __launch_bounds__(NW * 64) void attn_seq_x3_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                          const int* __restrict__ lens, int T, int H) {
  // K row stride = 8 mod 16 words: conflict-free ds_read_b128 fragment reads (gemm.hpp kGemmLdsStride); 288 keys only fit with + 4
  constexpr int C4 = HD / 4, KST = HD / 2 + (NKT <= 13 ? 8 : 4), NKB = (NKT + 1) / 2, VST = NKB * 16 + 4, NCH = HD / 32;
#if defined(MLDHIP_SIM)
  unsigned* KV = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_seq[];
  unsigned* KV = smem_seq;
#endif
  unsigned* Kh = KV;                         // phase 1: [NKT*16][KST] x 2 planes
  unsigned* Kl = KV + NKT * 16 * KST;
  unsigned* Vh = KV;                         // phase 2: [HD][VST] x 2 planes (V^T)
  unsigned* Vl = KV + HD * VST;
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  int len = T;
  if (lens) len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nkb = (nkt + 1) >> 1, nqt = (T + 15) >> 4;
  const int qt = blockIdx.y * NW + wave;
  const bool active = qt < nqt;                       // idle waves still take part in the barriers
  const float scale = rsqrtf((float)HD);

  // operand rows of this (sample, head) -> LDS planes; loads of a half-pass before its stores, clamped, rows >= len zeroed
  auto stage = [&](bool vphase) {
    constexpr int KPI = NW * 64 / C4, NIT = (NKB * 32 + KPI - 1) / KPI, UB = (NIT + 1) / 2;
    const int c4 = tid % C4, k0 = tid / C4;
    const float* base = qkv + (long long)b * T * 3 * D + (vphase ? 2 * D : D) + h * HD + c4 * 4;
#pragma unroll
    for (int j0 = 0; j0 < NIT; j0 += UB) {
      F4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int key = (j0 + u) * KPI + k0;
        const int kc = key < len ? key : len - 1;
        v[u] = ld4(base + (long long)kc * 3 * D);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int key = (j0 + u) * KPI + k0;
        const float m = key < len ? 1.f : 0.f;
        if (j0 + u < NIT) {
          unsigned h0, l0, h1, l1;
          split16_pair(v[u].x * m, v[u].y * m, h0, l0);
          split16_pair(v[u].z * m, v[u].w * m, h1, l1);
          if (!vphase) {
            if (key < nkt * 16) {
              *reinterpret_cast<U2*>(Kh + key * KST + c4 * 2) = U2{h0, h1};
              *reinterpret_cast<U2*>(Kl + key * KST + c4 * 2) = U2{l0, l1};
            }
          } else if (key < nkb * 32) {
            unsigned short* vh = reinterpret_cast<unsigned short*>(Vh) + key;
            unsigned short* vl = reinterpret_cast<unsigned short*>(Vl) + key;
            const int d0 = c4 * 4;
            vh[(d0 + 0) * VST * 2] = (unsigned short)(h0 & 0xFFFFu); vh[(d0 + 1) * VST * 2] = (unsigned short)(h0 >> 16);
            vh[(d0 + 2) * VST * 2] = (unsigned short)(h1 & 0xFFFFu); vh[(d0 + 3) * VST * 2] = (unsigned short)(h1 >> 16);
            vl[(d0 + 0) * VST * 2] = (unsigned short)(l0 & 0xFFFFu); vl[(d0 + 1) * VST * 2] = (unsigned short)(l0 >> 16);
            vl[(d0 + 2) * VST * 2] = (unsigned short)(l1 & 0xFFFFu); vl[(d0 + 3) * VST * 2] = (unsigned short)(l1 >> 16);
          }
        }
      }
    }
  };

  stage(false);
  __syncthreads();
  f32x4 s[2 * NKB];
#pragma unroll
  for (int kt = 0; kt < 2 * NKB; ++kt) s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float inv = 0.f;
  if (active) {
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = qkv + (long long)(b * T + qrow) * 3 * D + h * HD + g * 8;
    U4 qh[NCH], ql[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const F4 t0 = ld4(qp + c * 32), t1 = ld4(qp + c * 32 + 4);
      const float x[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
      split_hi_lo_x8(x, qh[c], ql[c]);
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const U4 kh = *reinterpret_cast<const U4*>(Kh + (kt * 16 + r) * KST + c * 16 + g * 4);
          const U4 kl = *reinterpret_cast<const U4*>(Kl + (kt * 16 + r) * KST + c * 16 + g * 4);
          s[kt] = mfma_x3_16x16x32(kl, qh[c], s[kt]);
          s[kt] = mfma_x3_16x16x32(kh, ql[c], s[kt]);
          s[kt] = mfma_x3_16x16x32(kh, qh[c], s[kt]);
        }
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2 * NKB; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (kt < nkt) && (kt * 16 + g * 4 + i < len);
        s[kt][i] = valid ? s[kt][i] : -INFINITY;
        m = fmaxf(m, s[kt][i]);
      }
    m = max_groups(m);
    float den = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2 * NKB; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = expf(s[kt][i] - m);
        s[kt][i] = e;
        den += e;
      }
    den = sum_groups(den);
    inv = 1.0f / den;
  }
  __syncthreads();                                    // every wave is done reading K
  stage(true);
  __syncthreads();
  if (!active) return;
  f32x4 oacc[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb < nkb) {
      const float pf[8] = {s[2 * kb][0] * inv, s[2 * kb][1] * inv, s[2 * kb][2] * inv, s[2 * kb][3] * inv,
                           s[2 * kb + 1][0] * inv, s[2 * kb + 1][1] * inv, s[2 * kb + 1][2] * inv, s[2 * kb + 1][3] * inv};
      U4 ph, pl;
      split_hi_lo_x8_unit(pf, ph, pl);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const unsigned* vh = Vh + (dt * 16 + r) * VST + kb * 16 + g * 2;
        const unsigned* vl = Vl + (dt * 16 + r) * VST + kb * 16 + g * 2;
        const U2 a0 = *reinterpret_cast<const U2*>(vh), a1 = *reinterpret_cast<const U2*>(vh + 8);
        const U2 b0 = *reinterpret_cast<const U2*>(vl), b1 = *reinterpret_cast<const U2*>(vl + 8);
        const U4 vhh = U4{a0.x, a0.y, a1.x, a1.y}, vll = U4{b0.x, b0.y, b1.x, b1.y};
        oacc[dt] = mfma_x3_16x16x32(pl, vhh, oacc[dt]);
        oacc[dt] = mfma_x3_16x16x32(ph, vll, oacc[dt]);
        oacc[dt] = mfma_x3_16x16x32(ph, vhh, oacc[dt]);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = qt * 16 + g * 4 + i;
      if (q < T) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[dt][i];
    }
}

// ----------------------------------------------------------------------------------------------
// Cross-attention of the trans_dec denoiser: every frame row attends to the TWO memory tokens [time, text]
// (mld_denoiser.py:170-172,214-215; cross_attention.py:336-339).  K/V of the time token are the same for all rows of
// a step (kv_time[2*D]: k then v), those of the text token are per sample (kv_text[sample][2*D]).
// One wave per row (D = 512: 8 contiguous dims per lane, a head = 16 lanes), 4 rows per workgroup.
template <int D, int HD>
__global__ __launch_bounds__(256) void cross2_kernel(const float* __restrict__ q, const float* __restrict__ kv_time,
                                                     const float* __restrict__ kv_text, float* __restrict__ o, int M, int T) {
  static_assert(D == 512 && HD == 128, "lane mapping below assumes 8 dims per lane and 16 lanes per head");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row = blockIdx.x * 4 + wave;
  const bool live = row < M;
  row = live ? row : M - 1;
  const int smp = row / T, c = lane * 8;
  const float scale = rsqrtf((float)HD);
  const F4 q0 = ld4(q + (long long)row * D + c), q1 = ld4(q + (long long)row * D + c + 4);
  const F4 a0 = ld4(kv_time + c), a1 = ld4(kv_time + c + 4);
  const F4 b0 = ld4(kv_text + (long long)smp * 2 * D + c), b1 = ld4(kv_text + (long long)smp * 2 * D + c + 4);
  const F4 va0 = ld4(kv_time + D + c), va1 = ld4(kv_time + D + c + 4);
  const F4 vb0 = ld4(kv_text + (long long)smp * 2 * D + D + c), vb1 = ld4(kv_text + (long long)smp * 2 * D + D + c + 4);
  const float qs[8] = {q0.x * scale, q0.y * scale, q0.z * scale, q0.w * scale, q1.x * scale, q1.y * scale, q1.z * scale, q1.w * scale};
  const float ka[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const float kb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { sa = fmaf(qs[i], ka[i], sa); sb = fmaf(qs[i], kb[i], sb); }
  sa = sum16(sa);
  sb = sum16(sb);
  const float m = fmaxf(sa, sb);
  const float ea = expf(sa - m), eb = expf(sb - m);
  const float inv = 1.0f / (ea + eb);
  const float pa = ea * inv, pb = eb * inv;
  if (!live) return;
  F4 y0, y1;
  y0.x = pa * va0.x + pb * vb0.x; y0.y = pa * va0.y + pb * vb0.y; y0.z = pa * va0.z + pb * vb0.z; y0.w = pa * va0.w + pb * vb0.w;
  y1.x = pa * va1.x + pb * vb1.x; y1.y = pa * va1.y + pb * vb1.y; y1.z = pa * va1.z + pb * vb1.z; y1.w = pa * va1.w + pb * vb1.w;
  st4(o + (long long)row * D + c, y0);
  st4(o + (long long)row * D + c + 4, y1);
}

// ----------------------------------------------------------------------------------------------
// The cross-attention sub-layer of the trans_dec denoiser WITHOUT its two GEMMs (round 6).  Its memory is two tokens per sample, so per head h
//   s_j = (x Wq_h^T + bq_h) . k_j / sqrt(128) = x . w_j + c_j        with  w_j = Wq_h^T k_j / sqrt(128)  (a 512-vector),  c_j = bq_h . k_j / sqrt(128)
//   out_proj(concat_h (p_1 v_1 + p_2 v_2)_h) = bo + sum_h (p_1^h u_1^h + p_2^h u_2^h)        with  u_j^h = Wo[:, head h] v_j^h  (a 512-vector)
// -- exact algebra (cross_attention.py:336-339, F.multi_head_attention_forward), the two-token form of the decoder's 1-key shortcut: the query projection and the
// out-projection (2 x 2 M 512^2 multiply-adds per layer, 26 GFLOP at M = 25 088) become 16 dot products / axpys of length 512 per row against vectors that depend on
// the memory tokens only.  cross_fold_kernel builds them: for the time token of every (layer, scheduler step) at finalize, for the text tokens once per call.
//   kv [L][ntok][2 D] (k | v; layer stride skv), Wq = in_proj_weight rows [0, D), bq, Wo = out_proj.weight [D][D] (layer stride sw) ->
//   w, u [L][ntok][H][D] (layer stride sfold), c [L][ntok][H] (layer stride sc).  grid = (ntok, H, L), block = 256.
constexpr int kCrossFoldLdsBytes = (2 * 128 + 4) * 4;
template <int D, int HD>
__global__ __launch_bounds__(256) void cross_fold_kernel(const float* __restrict__ kv, long long skv, const float* __restrict__ Wq, const float* __restrict__ bq,
                                                         const float* __restrict__ Wo, long long sw, float* __restrict__ w, float* __restrict__ u, long long sfold,
                                                         float* __restrict__ cc, long long sc, int ntok) {
  constexpr int H = D / HD;
#if defined(MLDHIP_SIM)
  float* sh = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float sh_fold[];      // kCrossFoldLdsBytes
  float* sh = sh_fold;
#endif
  float* ks = sh;            // [HD]
  float* vs = sh + HD;       // [HD]
  float* red = sh + 2 * HD;  // [4]
  const int tok = blockIdx.x, h = blockIdx.y, l = blockIdx.z, tid = threadIdx.x;
  const float* kvp = kv + (long long)l * skv + (long long)tok * 2 * D + h * HD;
  const float* wq = Wq + (long long)l * sw, *bqp = bq + (long long)l * sw, *wo = Wo + (long long)l * sw;
  const float scale = rsqrtf((float)HD);
  float part = 0.f;
  if (tid < HD) { const float k = kvp[tid]; ks[tid] = k; vs[tid] = kvp[D + tid]; part = bqp[h * HD + tid] * k; }
  part = sum64(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) cc[(long long)l * sc + (long long)tok * H + h] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
  float* wout = w + (long long)l * sfold + ((long long)tok * H + h) * D;
  float* uout = u + (long long)l * sfold + ((long long)tok * H + h) * D;
  for (int c = tid; c < D; c += 256) {
    float a = 0.f, b = 0.f;
    const float* wr = wo + (long long)c * D + h * HD;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
      a = fmaf(ks[d], wq[(long long)(h * HD + d) * D + c], a);      // coalesced over c
      b = fmaf(wr[d], vs[d], b);
    }
    wout[c] = a * scale;
    uout[c] = b;
  }
}

// [self-attention out-projection output Ha] + residual X -> LayerNorm 1 -> folded two-token cross-attention + residual -> LayerNorm 2 -> Y, one launch (it replaces
// add_layernorm, the query GEMM, cross2_kernel, the out-projection GEMM and the second add_layernorm of a trans_dec layer: mld_denoiser.py:208-221 via
// cross_attention.py:323-345).  A workgroup (4 waves) serves kC2Rows consecutive rows of ONE sample: the sample's sixteen folded vectors (8 of the time token, shared
// by every sample of the step, 8 of its text token) sit in LDS; a wave owns a row at a time, a lane the 8 columns c * 256 + 4 lane .. + 3 (add_layernorm_rows_kernel's map).
constexpr int kC2Rows = 28;
constexpr int kC2LdsBytes = (16 * 512 + 16 + 5 * 512) * 4;      // the sample's 16 folded vectors, 8 score constants, the five parameter rows: 43 KB
struct Cross2LnArgs {
  const float* Ha; const float* X;                 // [M][512] each: sub-layer output (bias included) and its residual
  const float* g1; const float* b1;                // LayerNorm 1
  const float* wt; const float* ut; const float* ct;      // time token of this (layer, step): [H][512], [H][512], [H]
  const float* wx; const float* ux; const float* cx;      // text tokens of this layer: [R][H][512], [R][H][512], [R][H]
  const float* bo;                                 // out_proj.bias [512]
  const float* g2; const float* b2;                // LayerNorm 2
  float* Y; int M, T;
  int src_mod = 0;                                 // > 0: Ha / X hold src_mod samples only and sample s reads sample s % src_mod's rows (layer 0 of a CFG batch: both halves share the rows in front of the first cross-attention)
};
// Eight per-lane partial sums -> the eight totals, in every lane: a reduce-scatter over the lane-swap instructions instead of eight 64-lane butterflies.
// v_permlane32_swap(a, b) exchanges a's upper 32 lanes with b's lower 32: a' + b' then holds the xor-32 pair sums of a in the lower half and of b in the upper half -- ONE
// swap + add folds TWO values; the same over 16-lane rows; the last four steps are DPP adds inside a row.  Row q of the two surviving registers holds the total of value
// (q & 1) * 4 + (q >> 1) resp. + 2: eight v_readlane bring them back as wave-uniform scalars.  28 instructions against 8 x 11.
__device__ __forceinline__ void sum64_x8(float (&v)[8]) {
#if defined(MLDHIP_SIM)
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = sum64(v[i]);
#else
  float a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float x = v[i], y = v[i + 4];
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    a[i] = x + y;                                  // lower half: pair sums of v[i]; upper half: of v[i + 4]
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float x = a[i], y = a[i + 2];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    b[i] = x + y;                                  // even rows: a[i]; odd rows: a[i + 2]
  }
  b[0] = sum16(b[0]);
  b[1] = sum16(b[1]);
  // rows (16-lane groups) 0..3 of b[i]: row 0 = lower half, even -> v[i]; row 1 = lower half, odd -> v[i + 2]; row 2 = upper half, even -> v[i + 4]; row 3 -> v[i + 6]
  const int b0 = __builtin_bit_cast(int, b[0]), b1 = __builtin_bit_cast(int, b[1]);
  v[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b0, 0));
  v[2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b0, 16));
  v[4] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b0, 32));
  v[6] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b0, 48));
  v[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b1, 0));
  v[3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b1, 16));
  v[5] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b1, 32));
  v[7] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b1, 48));
#endif
}

template <int D, int HD>
__global__ __launch_bounds__(256, 2) void cross2_fold_ln_kernel(Cross2LnArgs p) {
  static_assert(D == 512 && HD == 128, "four heads, two 256-column chunks per lane");
  constexpr int H = 4, NC = 2;
#if defined(MLDHIP_SIM)
  float* sm = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float sm_c2[];
  float* sm = sm_c2;
#endif
  float* Wl = sm;                  // [8][512]: w of (time, h = 0..3), (text, h = 0..3)
  float* Ul = sm + 8 * D;          // [8][512]: u likewise
  float* Cl = sm + 16 * D;         // [8]
  float* Pl = Cl + 16;             // [5][512]: LayerNorm 1 gain / bias, LayerNorm 2 gain / bias, out_proj.bias (read per row: 40 registers less, a workgroup more per CU)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (p.T + kC2Rows - 1) / kC2Rows;
  const int smp = blockIdx.x / per, r0 = (blockIdx.x % per) * kC2Rows;
  for (int i = tid; i < H * D / 4; i += 256) {
    st4(Wl + i * 4, ld4(p.wt + i * 4));
    st4(Wl + H * D + i * 4, ld4(p.wx + (long long)smp * H * D + i * 4));
    st4(Ul + i * 4, ld4(p.ut + i * 4));
    st4(Ul + H * D + i * 4, ld4(p.ux + (long long)smp * H * D + i * 4));
  }
  if (tid < D / 4) {
    st4(Pl + tid * 4, ld4(p.g1 + tid * 4)); st4(Pl + D + tid * 4, ld4(p.b1 + tid * 4));
    st4(Pl + 2 * D + tid * 4, ld4(p.g2 + tid * 4)); st4(Pl + 3 * D + tid * 4, ld4(p.b2 + tid * 4));
    st4(Pl + 4 * D + tid * 4, ld4(p.bo + tid * 4));
  }
  if (tid < H) { Cl[tid] = p.ct[tid]; Cl[H + tid] = p.cx[smp * H + tid]; }
  // the first row of this wave is in flight while the vectors are staged; inside the loop the NEXT row is requested before the current one is worked on
  auto row_of = [&](int t) { return (long long)smp * p.T + t; };
  const int ssrc = p.src_mod > 0 ? smp % p.src_mod : smp;
  auto src_of = [&](int t) { return (long long)ssrc * p.T + t; };
  F4 xa[NC], xr[NC];
  int t = r0 + wave;
  const int t_end = (r0 + kC2Rows < p.T ? r0 + kC2Rows : p.T);
  if (t < t_end) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { xa[c] = ld4(p.Ha + src_of(t) * D + c * 256 + lane * 4); xr[c] = ld4(p.X + src_of(t) * D + c * 256 + lane * 4); }
  }
  __syncthreads();
  auto layer_norm = [&](F4 (&x)[NC], const float* gm, const float* bt) __attribute__((always_inline)) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s += (x[c].x + x[c].y) + (x[c].z + x[c].w);
    const float mean = sum64(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      x[c].x -= mean; x[c].y -= mean; x[c].z -= mean; x[c].w -= mean;
      q += (x[c].x * x[c].x + x[c].y * x[c].y) + (x[c].z * x[c].z + x[c].w * x[c].w);
    }
    const float rstd = rsqrtf(sum64(q) * (1.0f / D) + kLnEps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const F4 g = ld4(gm + c * 256 + lane * 4), b = ld4(bt + c * 256 + lane * 4);
      x[c].x = x[c].x * rstd * g.x + b.x; x[c].y = x[c].y * rstd * g.y + b.y;
      x[c].z = x[c].z * rstd * g.z + b.z; x[c].w = x[c].w * rstd * g.w + b.w;
    }
  };
  for (; t < t_end; t += 4) {                                          // (wave-uniform bounds)
    const long long row = row_of(t);
    F4 x[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) x[c] = F4{xa[c].x + xr[c].x, xa[c].y + xr[c].y, xa[c].z + xr[c].z, xa[c].w + xr[c].w};
    if (t + 4 < t_end) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { xa[c] = ld4(p.Ha + src_of(t + 4) * D + c * 256 + lane * 4); xr[c] = ld4(p.X + src_of(t + 4) * D + c * 256 + lane * 4); }
    }
    layer_norm(x, Pl, Pl + D);                                         // h1 = LayerNorm1(x + self-attention)
    float sc[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const F4 wv = ld4(Wl + v * D + c * 256 + lane * 4);
        d += (x[c].x * wv.x + x[c].y * wv.y) + (x[c].z * wv.z + x[c].w * wv.w);
      }
      sc[v] = d;
    }
    sum64_x8(sc);
    float pa[H], pb[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float sa = sc[h] + Cl[h], sb = sc[H + h] + Cl[H + h];
      const float m = fmaxf(sa, sb);
      const float ea = expf(sa - m), eb = expf(sb - m);
      const float inv = 1.0f / (ea + eb);
      pa[h] = ea * inv; pb[h] = eb * inv;
    }
    F4 y[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      F4 o = ld4(Pl + 4 * D + c * 256 + lane * 4);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const F4 ua = ld4(Ul + h * D + c * 256 + lane * 4), ub = ld4(Ul + (H + h) * D + c * 256 + lane * 4);
        o.x += pa[h] * ua.x + pb[h] * ub.x; o.y += pa[h] * ua.y + pb[h] * ub.y;
        o.z += pa[h] * ua.z + pb[h] * ub.z; o.w += pa[h] * ua.w + pb[h] * ub.w;
      }
      y[c] = F4{x[c].x + o.x, x[c].y + o.y, x[c].z + o.z, x[c].w + o.w};
    }
    layer_norm(y, Pl + 2 * D, Pl + 3 * D);
#pragma unroll
    for (int c = 0; c < NC; ++c) st4(p.Y + row * D + c * 256 + lane * 4, y[c]);
  }
}

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) -> four N(0,1) draws per call through Box-Muller.  Counter-based, so the
// noise of (seed, step, element) does not depend on launch geometry: the in-engine replacement for the
// torch.randn(model_output.shape) inside DDPMScheduler.step (diffusers; call site mld.py:345-346).
struct Philox4 { unsigned v[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = unsigned(p1 >> 32) ^ c1 ^ k0, n1 = unsigned(p1), n2 = unsigned(p0 >> 32) ^ c3 ^ k1, n3 = unsigned(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float u01(unsigned x) { return float(x >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f); }   // (0, 1)
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned step, unsigned long long quad, float (&z)[4]) {
  const Philox4 p = philox4x32_10(unsigned(quad), unsigned(quad >> 32), step, 0u, unsigned(seed), unsigned(seed >> 32));
  const float r0 = sqrtf(-2.0f * logf(u01(p.v[0]))), t0 = 6.28318530717958647692f * u01(p.v[1]);
  const float r1 = sqrtf(-2.0f * logf(u01(p.v[2]))), t1 = 6.28318530717958647692f * u01(p.v[3]);
  z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
}

// out[n] = N(0,1) of (seed, step, element) -- exposed for tests / callers that want the engine's noise stream.
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, long long n, unsigned long long seed, unsigned step) {
  const long long nq = (n + 3) / 4;
  for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < nq; qd += (long long)gridDim.x * blockDim.x) {
    float z[4];
    philox_normal4(seed, step, (unsigned long long)qd, z);
    for (int i = 0; i < 4; ++i)
      if (qd * 4 + i < n) out[qd * 4 + i] = z[i];
  }
}

// DDPM ancestral step, variance_type fixed_small, epsilon prediction, no clipping (diffusers DDPMScheduler.step as
// restated in SURVEY.md App. A.3; third party, PARITY UNPINNED), with the classifier-free-guidance combine of
// mld.py:339-342 folded in when eps_cond != nullptr:
//   eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-ab_t) eps) / sqrt(ab_t);
//   x' = c_x0 x0 + c_x x + sigma * z,   z = noise[i] (injected) or Philox(seed, step, i); sigma = 0 at t = 0.
struct DdpmCoef { float sqrt_ab, sqrt_1mab, c_x0, c_x, sigma; };
__global__ __launch_bounds__(256) void cfg_ddpm_step_kernel(const float* __restrict__ eps_u, const float* __restrict__ eps_c,
                                                            const float* __restrict__ x, const float* __restrict__ noise,
                                                            float* __restrict__ out, long long n, float guidance, DdpmCoef k,
                                                            unsigned long long seed, unsigned step,
                                                            const unsigned long long* __restrict__ seed_ptr = nullptr) {
  if (seed_ptr) seed = *seed_ptr;        // captured step graphs take the call's seed from device memory (uniform load)
  const long long nq = (n + 3) / 4;
  for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < nq; qd += (long long)gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (k.sigma != 0.f && !noise) philox_normal4(seed, step, (unsigned long long)qd, z);
    for (int i = 0; i < 4; ++i) {
      const long long e = qd * 4 + i;
      if (e >= n) break;
      float ep = eps_u[e];
      if (eps_c) ep = ep + guidance * (eps_c[e] - ep);
      const float xv = x[e];
      const float x0 = (xv - k.sqrt_1mab * ep) / k.sqrt_ab;
      float y = k.c_x0 * x0 + k.c_x * xv;
      if (k.sigma != 0.f) y += k.sigma * (noise ? noise[e] : z[i]);
      out[e] = y;
    }
  }
}

}  // namespace mld
