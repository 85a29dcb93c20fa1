// Decoder self-attention kernel of the MLD engine (exact-fp32 MFMA path).  (The denoiser's 3-token
// attention lives in the A-prologue of tile32.hpp.)
//
// Replaces nn.MultiheadAttention's slow path as invoked by the reference
// (cross_attention.py:332-333 decoder self-attention): q pre-scaled by
// 1/sqrt(head_dim), scores, float -inf key-padding mask, softmax, P·V.  The [B·H, L, S] score
// tensor and the discarded head-averaged weights of the reference are never materialised.
//
// Input is the packed in-projection output qkv[row][3*D] (q | k | v, heads = contiguous 64-col
// slices); output o[row][D] feeds the out-proj GEMM.
#pragma once
#include "rt.hpp"

namespace mld {

// ----------------------------------------------------------------------------------------------
// VAE decoder self-attention over T frames with the key-padding mask of mld_vae.py:229
// (key t' of sample b masked iff t' >= len_b).  Rows are sample-major: row = b*T + t.
//
// One workgroup (4 waves) per (sample, head).  K and V of that (sample, head) -- T x 64 fp32 each,
// <= 2 x 56 KB -- are staged once in LDS (row stride 68 floats: conflict-free ds_read_b32 for V,
// <=2-way ds_read_b128 for K) and every wave walks 16-query tiles:
//   Sᵀ = K · Qᵀ (swapped operands): C tile col = query, row = key, so a lane owns ONE query column
//        and the softmax reductions are in-register + two xor-shuffles over the 4 row groups;
//   the P registers are then directly the A operand of P·V (k-slot g <-> key kt*16+4g+i) -- no
//   transpose, no LDS round trip for P (cdna_hip_programming.md T12 idea, fp32 form).
// Single pass (all scores of a 16-query tile live in registers: NKT tiles x 4 VGPRs), no online
// rescale needed at T <= 16*NKT.
// NW waves per workgroup: 8 puts two waves on every SIMD, so one wave's softmax (VALU) and LDS reads overlap the
// other's MFMAs -- with one wave per SIMD they serialise and the kernel runs at ~2.4x its MFMA time.
template <int NKT, int NW = 8>   // max key tiles (NKT*16 >= T); waves per workgroup
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                          const int* __restrict__ lens, int T, int H, int shared_qkv = 0) {
  constexpr int HD = 64, LDS_STRIDE = 68;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int bq = shared_qkv ? 0 : b;          // shared_qkv: every sample reads sample 0's projections (decoder layer 0: its input is the positional rows, the same for every sample)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4;          // key tiles that contain at least one valid key
  const int nqt = (len + 15) >> 4;          // query tiles: queries at padded frames are not computed (their rows are
                                            // never read by a valid row and the final layer zeroes them)
  float* Ks = smem;
  float* Vs = smem + (size_t)NKT * 16 * LDS_STRIDE;

  // ---- stage K and V.  All of a thread's global loads of one operand are issued before its first LDS store (a
  // load-store-per-iteration loop pays the memory latency once per iteration); addresses are clamped and rows >= len
  // zeroed by a multiply (P is 0 there and 0*garbage must not be NaN), so no load sits behind a branch.
  {
    constexpr int KPI = NW * 4, NIT = (NKT * 16 + KPI - 1) / KPI;   // keys per pass (16 float4 per 64-wide row); passes
    const int c4 = tid & 15, k0 = tid >> 4;
    const float* base = qkv + (long long)bq * T * 3 * D + h * HD + c4 * 4;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      float* dst = op == 0 ? Ks : Vs;
      const float* src = base + (op + 1) * D;
      F4 v[NIT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int key = j * KPI + k0;
        const int kc = key < len ? key : len - 1;
        v[j] = ld4(src + (long long)kc * 3 * D);
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int key = j * KPI + k0;
        const float m = key < len ? 1.f : 0.f;
        if (key < nkt * 16) st4(dst + key * LDS_STRIDE + c4 * 4, F4{v[j].x * m, v[j].y * m, v[j].z * m, v[j].w * m});
      }
    }
  }
  __syncthreads();

  for (int qt = wave; qt < nqt; qt += NW) {
    // Q fragment: query q0+r, head dims g*16 .. g*16+15, pre-scaled by 1/sqrt(64)
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = qkv + (long long)(bq * T + qrow) * 3 * D + h * HD + g * 16;
    float qf[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      F4 t = ld4(qp + c * 4);
      qf[c * 4] = t.x * 0.125f; qf[c * 4 + 1] = t.y * 0.125f; qf[c * 4 + 2] = t.z * 0.125f; qf[c * 4 + 3] = t.w * 0.125f;
    }
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        const float* kp = Ks + (kt * 16 + r) * LDS_STRIDE + g * 16;
        float kf[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          F4 t = ld4(kp + c * 4);
          kf[c * 4] = t.x; kf[c * 4 + 1] = t.y; kf[c * 4 + 2] = t.z; kf[c * 4 + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt] = mfma_f32_16x16x4(kf[i], qf[i], s[kt]);
      }
    }
    // masked softmax down each query column: this lane holds keys kt*16 + g*4 + i
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
        const float e = expf(s[kt][i] - m);   // exp(-inf) = 0 for masked keys
        s[kt][i] = e;
        den += e;
      }
    den = sum_groups(den);
    const float inv = 1.0f / den;
    // O = P · V
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pv = s[kt][i] * inv;
          const float* vp = Vs + (kt * 16 + g * 4 + i) * LDS_STRIDE + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[dt] = mfma_f32_16x16x4(pv, vp[dt * 16], oacc[dt]);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qt * 16 + g * 4 + i;
        if (q < T) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[dt][i];
      }
  }
}


// ----------------------------------------------------------------------------------------------
// Split-bf16 form of attn_decode_kernel (MLDHIP_PREC_BF16X3_DECODE and the other modes that run the decoder GEMMs on
// bf16 MFMAs): same one-workgroup-per-(sample, head) structure, same swapped QK^T / P-as-A-operand mapping, but every
// product runs as x = hi + lo in bf16 on v_mfma_f32_16x16x32_bf16 (3 MFMAs per 32-wide contraction chunk: lo*hi, hi*lo,
// hi*hi; fp32 accumulate) -- 6 + 6.5 MFMAs of 16 cycles per 16 x 16 score / 16 x 64 output tile pair instead of 16 + 16 fp32
// ones of 32 cycles.  Softmax, the 1/sqrt(d) scaling and the normalisation stay fp32.
//   K   in LDS as two bf16 planes [key][64 dims] (row stride 40 words): lane (r, g) reads dims 32c + 8g .. + 7 of key r.
//   V^T in LDS as two bf16 planes [dim][keys]: the P.V contraction runs over 32 KEYS per MFMA; a lane's 8 k-slots are the
//        keys it already holds scores for -- 16kt + 4g + {0..3} of the two key tiles (2kb, 2kb + 1) -- so P needs no shuffle,
//        and the matching V operand is two 8-byte reads of 4 consecutive keys each from the transposed planes.
constexpr int kAttnX3KStride = 40;   // words per K row: 64 bf16 = 32 words + 8 pad (conflict-free ds_read_b128 at word 16c + 4g of key r, gemm.hpp kGemmLdsStride)
template <int NKT>
constexpr int attn_x3_vt_stride() { return ((NKT + 1) / 2) * 16 + 4; }   // words per V^T row: an even number of key tiles of 16 bf16 (= 8 words) + pad
template <int NKT>
constexpr int attn_x3_lds_bytes() { return (2 * NKT * 16 * kAttnX3KStride + 2 * 64 * attn_x3_vt_stride<NKT>()) * 4; }

__device__ __forceinline__ void split_hi_lo_x8(const float (&x)[8], U4& hi, U4& lo) {
  split16_pair(x[0], x[1], hi.x, lo.x);
  split16_pair(x[2], x[3], hi.y, lo.y);
  split16_pair(x[4], x[5], hi.z, lo.z);
  split16_pair(x[6], x[7], hi.w, lo.w);
}

// the same for softmax numerators (bounded by construction -- [0, 1], or [0, 2^8] under the lazy reference of the key-blocked kernel: no range clamp)
__device__ __forceinline__ void split_hi_lo_x8_unit(const float (&x)[8], U4& hi, U4& lo) {
  split16_two(x[0], x[1], hi.x, lo.x);
  split16_two(x[2], x[3], hi.y, lo.y);
  split16_two(x[4], x[5], hi.z, lo.z);
  split16_two(x[6], x[7], hi.w, lo.w);
}

template <int NKT, int NW = 8>
__global__ __launch_bounds__(NW * 64) void attn_decode_x3_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                             const int* __restrict__ lens, int T, int H, int shared_qkv = 0) {
  constexpr int HD = 64, KST = kAttnX3KStride, VST = attn_x3_vt_stride<NKT>(), NKB = (NKT + 1) / 2;
#if defined(MLDHIP_SIM)
  unsigned* smem = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
  unsigned* smem = smem_u;
#endif
  unsigned* Kh = smem;                       // [NKT*16][KST]
  unsigned* Kl = Kh + NKT * 16 * KST;
  unsigned* Vh = Kl + NKT * 16 * KST;        // [64][VST]  (V^T: row = head dim, column = key)
  unsigned* Vl = Vh + 64 * VST;
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int bq = shared_qkv ? 0 : b;          // shared_qkv: every sample reads sample 0's projections (decoder layer 0: its input is the positional rows, the same for every sample)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nkb = (nkt + 1) >> 1, nqt = nkt;

  // ---- stage K (row-wise planes) and V (transposed planes); all global loads of an operand before its first LDS store,
  //      clamped addresses, rows >= len zeroed by a multiply (tile32 / attn_decode_kernel rules)
  {
    constexpr int KPI = NW * 4, NIT = (NKB * 32 + KPI - 1) / KPI;
    const int c4 = tid & 15, k0 = tid >> 4;
    const float* base = qkv + (long long)bq * T * 3 * D + h * HD + c4 * 4;
    F4 v[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int key = j * KPI + k0;
      const int kc = key < len ? key : len - 1;
      v[j] = ld4(base + D + (long long)kc * 3 * D);
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int key = j * KPI + k0;
      const float m = key < len ? 1.f : 0.f;
      if (key < nkt * 16) {
        unsigned h0, l0, h1, l1;
        split16_pair(v[j].x * m, v[j].y * m, h0, l0);
        split16_pair(v[j].z * m, v[j].w * m, h1, l1);
        *reinterpret_cast<U2*>(Kh + key * KST + c4 * 2) = U2{h0, h1};
        *reinterpret_cast<U2*>(Kl + key * KST + c4 * 2) = U2{l0, l1};
      }
    }
    // V^T: a thread owns one head dim and FOUR consecutive keys -- four coalesced 4-byte loads (a wave covers 256 contiguous bytes of
    // a key's row) and, after the split, ONE 8-byte LDS store per plane: the (key pair | key pair) words of its row.  (Owning four
    // dims of one key, as for K, means eight 2-byte stores per thread that land 8-way on the same banks: 4 x VST words apart.)
    constexpr int NVG = NKB * 8, NVI = (NVG + NW - 1) / NW;     // groups of 4 keys; iterations (a wave takes one group per iteration)
    const int vd = tid & 63, vg0 = tid >> 6;
    const float* vbase = qkv + (long long)bq * T * 3 * D + h * HD + 2 * D + vd;
    float vv[NVI][4];
#pragma unroll
    for (int it = 0; it < NVI; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int key = (it * NW + vg0) * 4 + q;
        const int kc = key < len ? key : len - 1;
        vv[it][q] = vbase[(long long)kc * 3 * D];
      }
#pragma unroll
    for (int it = 0; it < NVI; ++it) {
      const int vg = it * NW + vg0;
      if (vg < nkb * 8) {                    // keys of the (possibly half-empty) last 32-key block are zero
        float vm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) vm[q] = vg * 4 + q < len ? vv[it][q] : 0.f;
        unsigned h0, l0, h1, l1;
        split16_pair(vm[0], vm[1], h0, l0);
        split16_pair(vm[2], vm[3], h1, l1);
        *reinterpret_cast<U2*>(Vh + vd * VST + vg * 2) = U2{h0, h1};
        *reinterpret_cast<U2*>(Vl + vd * VST + vg * 2) = U2{l0, l1};
      }
    }
  }
  __syncthreads();

  for (int qt = wave; qt < nqt; qt += NW) {
    // Q fragments: query q0 + r, dims 32c + 8g .. + 7, pre-scaled by 1/sqrt(64), split hi / lo
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = qkv + (long long)(bq * T + qrow) * 3 * D + h * HD + g * 8;
    U4 qh[2], ql[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const F4 t0 = ld4(qp + c * 32), t1 = ld4(qp + c * 32 + 4);
      // 1/sqrt(64) and log2(e) in one factor: the scores come out in the log2 domain and the softmax runs on v_exp_f32 (2^x) directly
      constexpr float qs = 0.125f * 1.44269504088896340736f;
      const float x[8] = {t0.x * qs, t0.y * qs, t0.z * qs, t0.w * qs, t1.x * qs, t1.y * qs, t1.z * qs, t1.w * qs};
      split_hi_lo_x8(x, qh[c], ql[c]);
    }
    f32x4 s[2 * NKB];
#pragma unroll
    for (int kt = 0; kt < 2 * NKB; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < NKT && kt < nkt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const U4 kh = *reinterpret_cast<const U4*>(Kh + (kt * 16 + r) * KST + c * 16 + g * 4);
          const U4 kl = *reinterpret_cast<const U4*>(Kl + (kt * 16 + r) * KST + c * 16 + g * 4);
          s[kt] = mfma_x3_16x16x32(kl, qh[c], s[kt]);
          s[kt] = mfma_x3_16x16x32(kh, ql[c], s[kt]);
          s[kt] = mfma_x3_16x16x32(kh, qh[c], s[kt]);
        }
      }
    }
    // masked softmax down each query column: this lane holds keys kt*16 + g*4 + i
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
        const float e = fast_exp2(s[kt][i] - m);   // 2^(-inf) = 0 for masked keys
        s[kt][i] = e;
        den += e;
      }
    den = sum_groups(den);
    const float inv = 1.0f / den;
    // O = P . V over 32-key blocks: k-slot 8g + j <-> key (2kb + (j >> 2)) * 16 + 4g + (j & 3)
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < nkb) {
        const float pf[8] = {s[2 * kb][0] * inv, s[2 * kb][1] * inv, s[2 * kb][2] * inv, s[2 * kb][3] * inv,
                             s[2 * kb + 1][0] * inv, s[2 * kb + 1][1] * inv, s[2 * kb + 1][2] * inv, s[2 * kb + 1][3] * inv};
        U4 ph, pl;
        split_hi_lo_x8_unit(pf, ph, pl);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned* vh = Vh + (dt * 16 + r) * VST + kb * 16 + g * 2;     // keys 32kb + 4g .. + 3 (2 words), + 16 keys (8 words) for the second tile
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
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qt * 16 + g * 4 + i;
        if (q < T) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[dt][i];
      }
  }
}

// ----------------------------------------------------------------------------------------------
// Key-blocked (online-softmax) form of attn_decode_x3_kernel for head dim 64: the same split-bf16 products and operand layouts,
// but K and V pass through LDS in blocks of 32 keys (20 KB per stage, double buffered = 40 KB instead of 126 KB for a whole
// (sample, head)), so two workgroups share a CU and one's staging, softmax and store drain run under the other's MFMAs; inside a
// workgroup the next block's global loads are in flight while the current one is multiplied.  A wave works on BOTH of its query
// tiles (wave w: tiles w and w + 8) per key block, so every K / V fragment it reads from LDS feeds two tiles.  One kernel for every
// T (the block count is a run-time loop bound, uniform per workgroup).
//   running max m, running sum l and the unnormalised output O per query; per block: S = Q K_blk^T, m' = max(m, max S),
//   O *= exp(m - m'), P = exp(S - m'), l = l exp(m - m') + sum P, O += P V_blk; at the end O /= l.
// Scores of query r live in lanes (r, g = 0..3) (swapped QK^T, as above) while the accumulator rows of lane (r, g) are queries
// 4g + i: the per-query rescale factors are fetched with one lane shuffle per row (wave_bcast from lane 4g + i).
constexpr int kFlashKStride = 40;    // words per K row of a block: 64 bf16 = 32 words + 8 pad (conflict-free ds_read_b128, gemm.hpp)
constexpr int kFlashVStride = 20;    // words per V^T row of a block: 32 keys = 16 words + 4 pad (conflict-free ds_read_b64 at word 2g of row r)
constexpr int kFlashStageWords = 2 * 32 * kFlashKStride + 2 * 64 * kFlashVStride;
constexpr int kFlashLdsBytes = 2 * kFlashStageWords * 4;   // 40 960 B

// V is staged ROW-major like K ([key][64 dims] halves, 40-word rows, one 8-byte store per plane and thread) and the P V products read
// their B fragments -- four keys of one head dim per lane -- with ds_read_b64_tr_b16 (rt.hpp lds_read_tr16_b64).  (Rounds 2-3 wrote V^T
// planes with eight 2-byte stores per thread, 8-way bank conflicted: 454 vs 417 us per launch at 2 048 motions, same numbers to the bit;
// streaming hints on the loads / stores measured level.  Both alternatives were retired in round 4.)
__global__ __launch_bounds__(512, 4) void attn_flash_x3_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            const int* __restrict__ lens, int T, int H, int shared_qkv = 0) {
  constexpr int HD = 64, KST = kFlashKStride, VST = kFlashVStride, NW = 8;
#if defined(MLDHIP_SIM)
  unsigned* smem = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_flash[];
  unsigned* smem = smem_flash;
#endif
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int bq = shared_qkv ? 0 : b;          // shared_qkv: every sample reads sample 0's projections (decoder layer 0: its input is the positional rows, the same for every sample)
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nkb = (nkt + 1) >> 1, nqt = nkt;
  const float* base = qkv + (long long)bq * T * 3 * D + h * HD;

  // ---- staging: thread t owns key (t >> 4) of the block, dims 4 (t & 15) .. + 3 of K and of V
  const int skey = tid >> 4, c4 = tid & 15;
  F4 kreg, vreg;
  auto kvload = [&](int kb) {
    const int key = kb * 32 + skey;
    const int kc = key < len ? key : len - 1;
    const float* p = base + (long long)kc * 3 * D + c4 * 4;
    kreg = ld4(p + D);
    vreg = ld4(p + 2 * D);
  };
  auto kvstore = [&](int kb) {
    unsigned* Kh = smem + (kb & 1) * kFlashStageWords;
    unsigned* Kl = Kh + 32 * KST;
    unsigned* Vh = Kl + 32 * KST;
    unsigned* Vl = Vh + 64 * VST;
    const float m = kb * 32 + skey < len ? 1.f : 0.f;     // keys past the length: zero operands (their scores are masked as well)
    unsigned h0, l0, h1, l1;
    split16_pair(kreg.x * m, kreg.y * m, h0, l0);
    split16_pair(kreg.z * m, kreg.w * m, h1, l1);
    *reinterpret_cast<U2*>(Kh + skey * KST + c4 * 2) = U2{h0, h1};
    *reinterpret_cast<U2*>(Kl + skey * KST + c4 * 2) = U2{l0, l1};
    split16_pair(vreg.x * m, vreg.y * m, h0, l0);
    split16_pair(vreg.z * m, vreg.w * m, h1, l1);
    // row-major, K's row stride (the two V planes take 2 x 32 x 40 = 2 x 64 x 20 words)
    *reinterpret_cast<U2*>(Vh + skey * KST + c4 * 2) = U2{h0, h1};
    *reinterpret_cast<U2*>(Vl + skey * KST + c4 * 2) = U2{l0, l1};
  };
  kvload(0);

  // ---- this wave's query tiles: fragments of Q (pre-scaled by 1/sqrt(64), split hi / lo), running statistics, output accumulators
  bool live[2];
  U4 qh[2][2], ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = wave + NW * t;
    live[t] = qt < nqt;
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = base + (long long)qrow * 3 * D + g * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const F4 t0 = ld4(qp + c * 32), t1 = ld4(qp + c * 32 + 4);
      constexpr float qs = 0.125f * 1.44269504088896340736f;     // 1/sqrt(64) x log2(e): scores in the log2 domain (softmax on v_exp_f32)
      const float x[8] = {t0.x * qs, t0.y * qs, t0.z * qs, t0.w * qs, t1.x * qs, t1.y * qs, t1.z * qs, t1.w * qs};
      split_hi_lo_x8(x, qh[t][c], ql[t][c]);
    }
  }
  const int nt = live[1] ? 2 : 1;                  // waves 13 - 8 .. 7 of a 13-tile sequence have one tile: no MFMAs for a dead second one
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  f32x4 oacc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  kvstore(0);
  if (nkb > 1) kvload(1);
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const unsigned* Kh = smem + (kb & 1) * kFlashStageWords;
    const unsigned* Kl = Kh + 32 * KST;
    const unsigned* Vh = Kl + 32 * KST;
    const unsigned* Vl = Vh + 64 * VST;
    if (live[0]) {                                // waves without a query tile only stage (wave-uniform branch)
      // S = K_blk Q^T for the two 16-key tiles of the block, both query tiles
      f32x4 s[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) s[t][k2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const U4 kh = *reinterpret_cast<const U4*>(Kh + (k2 * 16 + r) * KST + c * 16 + g * 4);
          const U4 kl = *reinterpret_cast<const U4*>(Kl + (k2 * 16 + r) * KST + c * 16 + g * 4);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t < nt) {
              s[t][k2] = mfma_x3_16x16x32(kl, qh[t][c], s[t][k2]);
              s[t][k2] = mfma_x3_16x16x32(kh, ql[t][c], s[t][k2]);
              s[t][k2] = mfma_x3_16x16x32(kh, qh[t][c], s[t][k2]);
            }
          }
        }
      U4 ph[2], pl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= nt) continue;
        // online softmax of query r over this block's keys 32 kb + 16 k2 + 4 g + i, in the log2 domain.  The reference point m of a
        // query moves LAZILY: only when a block's maximum exceeds it by more than 8 (any reference is exact as long as 2^(s - m) stays
        // in range: here <= 2^8, far inside the half range of the split P), so after the first block or two the rescale of l and of
        // the 16 output accumulators -- and the four lane shuffles that fetch its factors -- is skipped by a wave-uniform branch.
        if (kb * 32 + 32 > len) {                 // the block that crosses the length: keys past it get -inf (wave-uniform test)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[t][k2][i] = kb * 32 + k2 * 16 + g * 4 + i < len ? s[t][k2][i] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3])),
                         fmaxf(fmaxf(s[t][1][0], s[t][1][1]), fmaxf(s[t][1][2], s[t][1][3])));
        mx = max_groups(mx);                       // finite: key 32 kb < len
        if (wave_any(mx > mrun[t] + 8.0f)) {       // first block: mrun = -inf
          const float mnew = fmaxf(mrun[t], mx);
          const float alpha = fast_exp2(mrun[t] - mnew);   // 2^(-inf) = 0 on the first block
          lrun[t] *= alpha;
          mrun[t] = mnew;
          // accumulator row i of this lane is query 4 g + i: its rescale factor sits in lane 4 g + i
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float a = wave_bcast(alpha, g * 4 + i);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[t][dt][i] *= a;
          }
        }
        float psum = 0.f;
        float pf[8];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = fast_exp2(s[t][k2][i] - mrun[t]);   // 2^(-inf) = 0 for masked keys
            pf[k2 * 4 + i] = e;
            psum += e;
          }
        lrun[t] += sum_groups(psum);
        split_hi_lo_x8_unit(pf, ph[t], pl[t]);
      }
      // O += P V_blk: k-slot 8 g + j <-> key (j >> 2) * 16 + 4 g + (j & 3) of the block
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        U2 a0, a1, b0, b1;
        // lane (r, g) supplies row = key 4 g + (r >> 2) of the 16-key tile, dims 16 dt + 4 (r & 3) .. + 3, and receives dim 16 dt + r of
        // keys 4 g .. 4 g + 3 (8-byte aligned: even word offsets; rows 0 .. 7 of a 32-lane half sit 40 words apart: all 64 banks once)
        const int tro = (g * 4 + (r >> 2)) * KST + dt * 8 + (r & 3) * 2;
        a0 = lds_read_tr16_b64(Vh + tro); a1 = lds_read_tr16_b64(Vh + tro + 16 * KST);
        b0 = lds_read_tr16_b64(Vl + tro); b1 = lds_read_tr16_b64(Vl + tro + 16 * KST);
        const U4 vhh = U4{a0.x, a0.y, a1.x, a1.y}, vll = U4{b0.x, b0.y, b1.x, b1.y};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t < nt) {
            oacc[t][dt] = mfma_x3_16x16x32(pl[t], vhh, oacc[t][dt]);
            oacc[t][dt] = mfma_x3_16x16x32(ph[t], vll, oacc[t][dt]);
            oacc[t][dt] = mfma_x3_16x16x32(ph[t], vhh, oacc[t][dt]);
          }
        }
      }
    }
    if (kb + 1 < nkb) {
      kvstore(kb + 1);
      if (kb + 2 < nkb) kvload(kb + 2);
    }
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) continue;
    const int qt = wave + NW * t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float inv = 1.0f / wave_bcast(lrun[t], g * 4 + i);
      const int q = qt * 16 + g * 4 + i;
      if (q < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[t][dt][i] * inv;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// The key-blocked kernel for head dim 128 (the diffusion-only variant's trans_dec denoiser: 4 heads x 128, T = 196, no key-padding
// mask: mld_denoiser.py:208-221, cross_attention.py:332-333).  Same algorithm, operand layouts and lane maps as attn_flash_x3_kernel --
// blocks of 32 keys double buffered in LDS, a wave works on its two query tiles (w, w + 8) per block so every K / V fragment feeds two
// tiles, lazy reference point, V through the transpose read -- at twice the row width: 4 contraction chunks for Q K^T, 8 output dim
// tiles, 72-word rows (= 8 mod 16), 36.9 KB per stage.  Two query tiles x (8 + 8) fragments + 64 output accumulators per lane need the
// 256-register budget: one workgroup per CU, two waves per SIMD (with ONE tile per wave the fragment reads per matrix instruction
// double and LDS, not the matrix pipe, is the bound).  Replaces attn_seq_x3_kernel (novae.hpp: K, then V, through one LDS buffer,
// 141 us per launch = 71 TFLOP/s at the BASELINE shape) when T <= 256.  lens == nullptr: every key counts.
constexpr int kFlash128KStride = 72;      // words per K / V row of a block: 128 halves = 64 words + 8 pad
constexpr int kFlash128StageWords = 4 * 32 * kFlash128KStride;        // K high / low, V high / low
constexpr int kFlash128LdsBytes = 2 * kFlash128StageWords * 4;        // 73 728 B

__global__ __launch_bounds__(512, 2) void attn_flash128_x3_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                               const int* __restrict__ lens, int T, int H) {
  constexpr int HD = 128, KST = kFlash128KStride, NW = 8, NCH = HD / 32, NDT = HD / 16;
#if defined(MLDHIP_SIM)
  unsigned* smem = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_flash128[];
  unsigned* smem = smem_flash128;
#endif
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  int len = T;
  if (lens) len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nkb = (nkt + 1) >> 1, nqt = (T + 15) >> 4;
  const float* base = qkv + (long long)b * T * 3 * D + h * HD;

  // ---- staging: thread t owns dims 4 (t & 31) .. + 3 of keys (t >> 5) and 16 + (t >> 5) of the block, K and V
  const int skey = tid >> 5, c4 = tid & 31;
  F4 kreg[2], vreg[2];
  auto kvload = [&](int kb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = kb * 32 + u * 16 + skey;
      const int kc = key < len ? key : len - 1;
      const float* p = base + (long long)kc * 3 * D + c4 * 4;
      kreg[u] = ld4(p + D);
      vreg[u] = ld4(p + 2 * D);
    }
  };
  auto kvstore = [&](int kb) {
    unsigned* Kh = smem + (kb & 1) * kFlash128StageWords;
    unsigned* Kl = Kh + 32 * KST;
    unsigned* Vh = Kl + 32 * KST;
    unsigned* Vl = Vh + 32 * KST;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = u * 16 + skey;
      const float m = kb * 32 + row < len ? 1.f : 0.f;     // keys past the length: zero operands (their scores are masked as well)
      unsigned h0, l0, h1, l1;
      split16_pair(kreg[u].x * m, kreg[u].y * m, h0, l0);
      split16_pair(kreg[u].z * m, kreg[u].w * m, h1, l1);
      *reinterpret_cast<U2*>(Kh + row * KST + c4 * 2) = U2{h0, h1};
      *reinterpret_cast<U2*>(Kl + row * KST + c4 * 2) = U2{l0, l1};
      split16_pair(vreg[u].x * m, vreg[u].y * m, h0, l0);
      split16_pair(vreg[u].z * m, vreg[u].w * m, h1, l1);
      *reinterpret_cast<U2*>(Vh + row * KST + c4 * 2) = U2{h0, h1};
      *reinterpret_cast<U2*>(Vl + row * KST + c4 * 2) = U2{l0, l1};
    }
  };
  kvload(0);

  // ---- this wave's query tiles: fragments of Q (pre-scaled by 1/sqrt(128) x log2(e), split hi / lo), running statistics, accumulators
  bool live[2];
  U4 qh[2][NCH], ql[2][NCH];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = wave + NW * t;
    live[t] = qt < nqt;
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const float* qp = base + (long long)qrow * 3 * D + g * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const F4 t0 = ld4(qp + c * 32), t1 = ld4(qp + c * 32 + 4);
      constexpr float qs = 0.08838834764831845f * 1.44269504088896340736f;     // 1/sqrt(128) x log2(e)
      const float x[8] = {t0.x * qs, t0.y * qs, t0.z * qs, t0.w * qs, t1.x * qs, t1.y * qs, t1.z * qs, t1.w * qs};
      split_hi_lo_x8(x, qh[t][c], ql[t][c]);
    }
  }
  const int nt = live[1] ? 2 : 1;
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  f32x4 oacc[2][NDT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  kvstore(0);
  if (nkb > 1) kvload(1);
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const unsigned* Kh = smem + (kb & 1) * kFlash128StageWords;
    const unsigned* Kl = Kh + 32 * KST;
    const unsigned* Vh = Kl + 32 * KST;
    const unsigned* Vl = Vh + 32 * KST;
    if (live[0]) {                                // waves without a query tile only stage (wave-uniform branch)
      f32x4 s[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) s[t][k2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const U4 kh = *reinterpret_cast<const U4*>(Kh + (k2 * 16 + r) * KST + c * 16 + g * 4);
          const U4 kl = *reinterpret_cast<const U4*>(Kl + (k2 * 16 + r) * KST + c * 16 + g * 4);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t < nt) {
              s[t][k2] = mfma_x3_16x16x32(kl, qh[t][c], s[t][k2]);
              s[t][k2] = mfma_x3_16x16x32(kh, ql[t][c], s[t][k2]);
              s[t][k2] = mfma_x3_16x16x32(kh, qh[t][c], s[t][k2]);
            }
          }
        }
      U4 ph[2], pl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= nt) continue;
        if (kb * 32 + 32 > len) {                 // the block that crosses the length: keys past it get -inf (wave-uniform test)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[t][k2][i] = kb * 32 + k2 * 16 + g * 4 + i < len ? s[t][k2][i] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3])),
                         fmaxf(fmaxf(s[t][1][0], s[t][1][1]), fmaxf(s[t][1][2], s[t][1][3])));
        mx = max_groups(mx);                       // finite: key 32 kb < len
        if (wave_any(mx > mrun[t] + 8.0f)) {       // lazy reference point (attn_flash_x3_kernel); first block: mrun = -inf
          const float mnew = fmaxf(mrun[t], mx);
          const float alpha = fast_exp2(mrun[t] - mnew);
          lrun[t] *= alpha;
          mrun[t] = mnew;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float a = wave_bcast(alpha, g * 4 + i);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) oacc[t][dt][i] *= a;
          }
        }
        float psum = 0.f;
        float pf[8];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = fast_exp2(s[t][k2][i] - mrun[t]);
            pf[k2 * 4 + i] = e;
            psum += e;
          }
        lrun[t] += sum_groups(psum);
        split_hi_lo_x8_unit(pf, ph[t], pl[t]);
      }
      // O += P V_blk through the transpose read: lane (r, g) supplies key 4 g + (r >> 2) of a 16-key tile, dims 16 dt + 4 (r & 3) .. + 3
      // (rows 72 = 8 mod 64 words apart: the eight rows of a 32-lane half cover all 64 banks once)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int tro = (g * 4 + (r >> 2)) * KST + dt * 8 + (r & 3) * 2;
        const U2 a0 = lds_read_tr16_b64(Vh + tro), a1 = lds_read_tr16_b64(Vh + tro + 16 * KST);
        const U2 b0 = lds_read_tr16_b64(Vl + tro), b1 = lds_read_tr16_b64(Vl + tro + 16 * KST);
        const U4 vhh = U4{a0.x, a0.y, a1.x, a1.y}, vll = U4{b0.x, b0.y, b1.x, b1.y};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t < nt) {
            oacc[t][dt] = mfma_x3_16x16x32(pl[t], vhh, oacc[t][dt]);
            oacc[t][dt] = mfma_x3_16x16x32(ph[t], vll, oacc[t][dt]);
            oacc[t][dt] = mfma_x3_16x16x32(ph[t], vhh, oacc[t][dt]);
          }
        }
      }
    }
    if (kb + 1 < nkb) {
      kvstore(kb + 1);
      if (kb + 2 < nkb) kvload(kb + 2);
    }
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) continue;
    const int qt = wave + NW * t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float inv = 1.0f / wave_bcast(lrun[t], g * 4 + i);
      const int q = qt * 16 + g * 4 + i;
      if (q < T) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[(long long)(b * T + q) * D + h * HD + dt * 16 + r] = oacc[t][dt][i] * inv;
      }
    }
  }
}

}  // namespace mld
