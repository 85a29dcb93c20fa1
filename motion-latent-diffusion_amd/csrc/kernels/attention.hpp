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
                                                          const int* __restrict__ lens, int T, int H) {
  constexpr int HD = 64, LDS_STRIDE = 68;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  const int D = H * HD;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
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
    const float* base = qkv + (long long)b * T * 3 * D + h * HD + c4 * 4;
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
    const float* qp = qkv + (long long)(b * T + qrow) * 3 * D + h * HD + g * 16;
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

}  // namespace mld
