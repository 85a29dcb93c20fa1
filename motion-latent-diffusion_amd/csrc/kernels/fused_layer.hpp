// Fused sub-layer kernels of the denoiser loop: fewer, fatter launches.  MEASURED NEGATIVE RESULT, kept as an A/B knob
// (MLDHIP_FUSED_FFN=1, default off): the fused FFN takes 10.8 us against 6.9 + 4.8 us for the two tile32 launches it
// replaces, and its 8 partial slabs cost the consumers +1.1 us (QKV prologue) and +0.8 us (step-final kernel): 3 808
// vs 3 840 motions/s (profiles/r01_v13_fused_ffn_ab.txt).  One workgroup per CU runs load -> LDS -> MFMA -> LDS ->
// MFMA -> store strictly in sequence, and at 16 rows x 128 hidden columns the two fp32-MFMA phases alone are 3.5 us;
// two separate launches overlap those phases across workgroups for the price of one ~1.5 us boundary.  Same verdict as
// cdna_hip_programming.md §5.6 ("M = 256 residual block: cut at every seam").
//
// Why: at M = 6B = 384 rows every launch of the reverse loop costs ~1.5 us of dependent-kernel boundary plus
// ~1.5-2 us of exposed memory latency before its first MFMA (profiles/r01_v7_tile32_phase_trace.json), and a
// post-norm encoder layer is a chain of four GEMMs.  GEMM -> GEMM fusion is possible without any inter-workgroup
// hand-off when the K dimension of the second GEMM is split over workgroups: a workgroup that owns
// (16 rows) x (a 128-wide slice of the FFN hidden dimension) computes its slice of gelu(h W1^T + b1) and immediately
// multiplies it by the matching 128 columns of W2 -- the result is a raw split-K partial slab, exactly what the
// consumer prologues of tile32.hpp already sum.  Weights never depend on activations, so all of a workgroup's global
// loads (A rows, the W1 slice AND the W2 slice) are issued up front; the W2 slice waits in registers while phase 1 runs.
#pragma once
#include "tile32.hpp"

namespace mld {

struct FfnFusedArgs {
  ASrc src;                      // combine source of the FFN input rows h1 = LN1(out-proj slabs + bias + residual)
  const float* W1 = nullptr;     // linear1.weight [F][256]
  const float* b1 = nullptr;     // linear1.bias [F]
  const float* W2 = nullptr;     // linear2.weight [256][F]
  float* P = nullptr;            // raw FFN2 partial slabs [F/128][M][256]
  long long pstride = 0;
  int M = 0, F = 0;
};

constexpr int kFfnHS = 128;                                   // hidden columns per workgroup
constexpr int kFfnLdsFloats = 16 * kT32Stride + kFfnHS * kT32Stride;   // A tile + W1 slice (phase 2 overlays it)
constexpr int kFfnLdsBytes = kFfnLdsFloats * 4;              // 149,760 B -> one workgroup per CU

// Wave-local assembly of RPW rows of a combine source (sum of NS slabs + bias [+ residual] [+ LayerNorm]); lane l owns
// columns 4l..4l+3.  All loads are issued before the first dependent instruction.
template <int RPW, int NS>
__device__ __forceinline__ void assemble_rows(const ASrc& src, const int (&rows)[RPW], int lane, F4 (&areg)[RPW]) {
  F4 sl[RPW][NS], rs[RPW];
  const bool has_res = src.res != nullptr, has_ln = src.gamma != nullptr;
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int s = 0; s < NS; ++s) sl[i][s] = ld4(src.base + s * src.pstride + (long long)rows[i] * 256 + lane * 4);
  if (has_res) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) rs[i] = ld4(src.res + (long long)rows[i] * src.ldres + lane * 4);
  }
  const F4 bias = ld4(src.bias + lane * 4);
  F4 gm = F4{1.f, 1.f, 1.f, 1.f}, bt = F4{0.f, 0.f, 0.f, 0.f};
  if (has_ln) { gm = ld4(src.gamma + lane * 4); bt = ld4(src.beta + lane * 4); }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {   // ((s0+s1)+s2)+..., + bias, + res  (same association as tile32's prologue)
    F4 v = sl[i][0];
#pragma unroll
    for (int s = 1; s < NS; ++s) v = f4add(v, sl[i][s]);
    v = f4add(v, bias);
    if (has_res) v = f4add(v, rs[i]);
    areg[i] = v;
  }
  if (has_ln) {
    float s[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) s[i] = areg[i].x + areg[i].y + areg[i].z + areg[i].w;
    sum64xn<RPW>(s);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float mean = s[i] * (1.0f / 256.0f);
      areg[i] = F4{areg[i].x - mean, areg[i].y - mean, areg[i].z - mean, areg[i].w - mean};
      s[i] = areg[i].x * areg[i].x + areg[i].y * areg[i].y + areg[i].z * areg[i].z + areg[i].w * areg[i].w;
    }
    sum64xn<RPW>(s);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float rstd = rsqrtf(s[i] * (1.0f / 256.0f) + kLnEps);
      areg[i] = F4{areg[i].x * rstd * gm.x + bt.x, areg[i].y * rstd * gm.y + bt.y, areg[i].z * rstd * gm.z + bt.z,
                   areg[i].w * rstd * gm.w + bt.w};
    }
  }
}

// 64 MFMAs: one 16x16 output tile over K = 256 (two accumulators), fragments from LDS rows of stride `st`.
template <int KCH>
__device__ __forceinline__ f32x4 tile_mfma(const float* ap, const float* wp) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < KCH; ++kc) {
    const F4 a0 = ld4(ap + kc * 32), a1 = ld4(ap + kc * 32 + 4);
    const F4 b0 = ld4(wp + kc * 32), b1 = ld4(wp + kc * 32 + 4);
    acc0 = mfma_f32_16x16x4(a0.x, b0.x, acc0);
    acc1 = mfma_f32_16x16x4(a0.y, b0.y, acc1);
    acc0 = mfma_f32_16x16x4(a0.z, b0.z, acc0);
    acc1 = mfma_f32_16x16x4(a0.w, b0.w, acc1);
    acc0 = mfma_f32_16x16x4(a1.x, b1.x, acc0);
    acc1 = mfma_f32_16x16x4(a1.y, b1.y, acc1);
    acc0 = mfma_f32_16x16x4(a1.z, b1.z, acc0);
    acc1 = mfma_f32_16x16x4(a1.w, b1.w, acc1);
  }
  return acc0 + acc1;
}

// FFN of one post-norm encoder layer (cross_attention.py:268-271): P[z] = gelu(h1 W1[z]^T + b1[z]) W2[:, z]^T for the
// 128-wide hidden slice z = blockIdx.y and the 16 rows blockIdx.x; h1 is assembled on load (norm1 of the out-proj
// slabs + bias + residual) and written back by the z == 0 workgroups (it is the FFN residual of the consumer).
// grid = (ceil(M/16), F/128), block = 512 (8 waves), dynamic LDS = kFfnLdsBytes.
template <int NS0>
__global__ __launch_bounds__(512) void den_ffn_fused_kernel(FfnFusedArgs p) {
  constexpr int ST = kT32Stride, HST = kFfnHS + 4;       // LDS strides: 260 (K = 256 rows), 132 (K = 128 rows)
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* As = smem;                    // phase 1: A tile [16][260] | W1 slice [128][260]
  float* Ws = smem + 16 * ST;
  float* Hs = smem;                    // phase 2: hidden tile [16][132] | W2 slice [256][132]   (overlay)
  float* W2s = smem + 16 * HST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 16, z = blockIdx.y;
  const int r = lane & 15, g = lane >> 4;

  // ---- every global load of the workgroup, up front
  F4 w1reg[16], w2reg[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w1reg[i] = ld4(p.W1 + (long long)(z * kFfnHS + wave + i * 8) * 256 + lane * 4);
  const int o2 = (lane >> 5), c2 = (lane & 31) * 4;       // W2 slice: two 128-float row pieces per wave instruction
#pragma unroll
  for (int i = 0; i < 16; ++i) w2reg[i] = ld4(p.W2 + (long long)((i * 8 + wave) * 2 + o2) * p.F + z * kFfnHS + c2);
  const float ebias = p.b1[z * kFfnHS + wave * 16 + r];
  int rows[2];
  bool live[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wave + i * 8;
    live[i] = row < p.M;
    rows[i] = live[i] ? row : p.M - 1;
  }
  F4 areg[2];
  assemble_rows<2, NS0>(p.src, rows, lane, areg);
  if (p.src.out && z == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (live[i]) st4(p.src.out + (long long)rows[i] * p.src.ldout + lane * 4, areg[i]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) st4(Ws + (wave + i * 8) * ST + lane * 4, w1reg[i]);
#pragma unroll
  for (int i = 0; i < 2; ++i) st4(As + (wave + i * 8) * ST + lane * 4, areg[i]);
  __syncthreads();

  // ---- phase 1: hidden[16][128] = gelu(A W1^T + b1); wave w owns hidden columns 16w..16w+15
  f32x4 h = tile_mfma<8>(As + r * ST + g * 8, Ws + (wave * 16 + r) * ST + g * 8);
  __syncthreads();                                        // A tile and W1 slice are dead: overlay them
#pragma unroll
  for (int i = 0; i < 4; ++i) Hs[(g * 4 + i) * HST + wave * 16 + r] = gelu_erf(h[i] + ebias);
#pragma unroll
  for (int i = 0; i < 16; ++i) st4(W2s + ((i * 8 + wave) * 2 + o2) * HST + c2, w2reg[i]);
  __syncthreads();

  // ---- phase 2: partial[16][256] = hidden W2[:, slice]^T over K = 128; wave w owns output column tiles w and w + 8
  float* P = p.P + z * p.pstride;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ct = wave + t * 8;
    const f32x4 acc = tile_mfma<4>(Hs + r * HST + g * 8, W2s + (ct * 16 + r) * HST + g * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + g * 4 + i;
      if (row < p.M) P[(long long)row * 256 + ct * 16 + r] = acc[i];
    }
  }
}

}  // namespace mld
