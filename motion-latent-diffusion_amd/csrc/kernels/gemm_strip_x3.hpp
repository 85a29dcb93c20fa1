// Row-strip GEMM of the decoder / encoder layers, split-f16 operands, weights register-direct (round 3):
//   Y[M, N] = A[M, K] W^T + bias                                  (in-projection N = 768; skip linear K = 512 as two A segments)
//   Y[M, 256] = LN2( LN1(A W^T + bias + res) + cvec[sample] )      (self-attention out-projection + residual + norm1, then the
//                                                                   1-key cross-attention vector + norm2: gemm.hpp's LN epilogue)
// Same structure as ffn_strip.hpp: a workgroup keeps RT x 16 rows of A in LDS as a split image, wave w owns columns 16w .. 16w + 15
// of every 128-column block and loads its MFMA operands of the weights straight from a fragment-ordered stream (`finalize`
// re-packs each weight: per pair of column blocks, per K segment, per chunk, [block 2p, block 2p + 1]) into a register ring; the two
// blocks of a pair share the A fragments.  One strip per workgroup instead of one 64 x 128 tile: A is read from HBM / L2 once per
// strip (not once per column tile), weights once per 96 rows (not per 64), no barrier per K chunk.  Output tiles leave through LDS
// with 16-byte stores.  Replaces gemm_kernel<2,4,2,2,...,STAGED,X3> / <2,4,2,4,LN,...> of the split modes (cross_attention.py:323-339).
#pragma once
#include "ffn_strip.hpp"

namespace mld {

struct StripGemmArgs {
  const float* A = nullptr;          // [M][256] first K segment
  const float* A2 = nullptr;         // [M][256] second K segment (skip concat) or NULL
  const float* W = nullptr;          // fragment-ordered stream of the weight (N / 256 pairs x NSEG x 8 chunks x 2 items)
  const float* bias = nullptr;       // [N]
  float* Y = nullptr; int ldy = 0;
  int M = 0, N = 0;                  // N % 256 == 0
  const int* skip_lens = nullptr; int skip_rpg = 1;      // skip strips made only of padded frames
  // LN form (N == 256)
  const float* res = nullptr;        // [M][256]
  const float* g1 = nullptr; const float* b1 = nullptr;
  const float* cvec = nullptr; int rpg = 1;              // + cvec[row / rpg][256] before the second LayerNorm
  const float* g2 = nullptr; const float* b2 = nullptr;
};

#ifndef SB_EXP
#define SB_EXP 0          // tools/loopbench/strip_bench.hip experiments (measurement only, 0 in the library)
#endif
template <int RT, int NSEG, bool STAGE>
constexpr int strip_gemm_lds_bytes() { return (NSEG * RT * 16 * kFsXs + (STAGE ? RT * 16 * kFsHs : 0) + 2 * 8 * RT * 16 + RT * 16) * 4; }

// grid = ceil(M / (16 RT)); block = 512.  STAGE: a separate [rows][136] staging tile for the output (needed when A must survive
// the first pair: N > 256); otherwise the output is parked in A's own rows once the last product is done.
// NT: the strip's loads and the output stores carry the streaming hint -- activations that exactly one
// workgroup touches once per launch, next to weight streams every workgroup re-reads.  A COMPILE-time choice: as a run-time branch
// around each access (the first form, r03_late_options_ab.json) it made the whole kernel 12-30 % slower -- a `cond ? load : load`
// per element drains the memory counter per access (DESIGN.md point 8).
// (Round 3 also carried a run-time-selectable ring depth and separate load / store hint bits; round 4 keeps the measured forms only: 8 items in
// flight per lane -- 4 in the LayerNorm form, whose epilogue needs the registers -- and the hinted build for the N = 768 in-projection.)
template <int RT, int NSEG, bool LN, bool STAGE, bool NT = false>
__global__ __launch_bounds__(512, (RT * NSEG <= 4 && RT <= 3 ? 4 : 2)) void strip_gemm_x3_kernel(StripGemmArgs p) {      // RT <= 3: 80 KB of LDS and 128 registers -- two workgroups per CU
  static_assert(!(LN && (NSEG != 1 || STAGE)), "the LayerNorm form is the N = 256, K = 256 out-projection");
#ifndef SB_RING3
#define SB_RING3 4        // items in flight per lane of the 48-row form (128 registers: 8 spill 40 B)
#endif
  constexpr int RING = LN ? 4 : (RT <= 3 ? SB_RING3 : 8);   // (RT <= 3 with one K segment, RT = 2 with two: the forms at four waves per SIMD)      // weight items in flight per lane
  constexpr int BM = RT * 16, XS = kFsXs, HS = kFsHs;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem_sg[];
  float* smem = smem_sg;
#endif
  float* Xs = smem;                                  // [NSEG][BM][264] split images of the strip's K segments
  float* St = Xs + NSEG * BM * XS;                   // [BM][136] output staging (STAGE)
  float* red = St + (STAGE ? BM * HS : 0);           // [2][8][BM] LayerNorm partial sums
  int* sidx = reinterpret_cast<int*>(red + 2 * 8 * BM);   // [BM] sample of each row (LN form: which cvec row to add)
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int col0 = wave * 16 + r;

  if (p.skip_lens) {                                 // uniform exit for strips of padded frames only (gemm.hpp)
    const int t0 = m0, t1 = (t0 + BM < p.M ? t0 + BM : p.M) - 1;
    bool all_padding = true;
    for (int b = t0 / p.skip_rpg; b <= t1 / p.skip_rpg; ++b) {
      const int first = (t0 > b * p.skip_rpg ? t0 : b * p.skip_rpg) - b * p.skip_rpg;
      if (first < p.skip_lens[b]) { all_padding = false; break; }
    }
    if (all_padding) return;
  }

  const int npairs = p.N >> 8, nitems = npairs * NSEG * 16;
  const float* gsrc = p.W + tid * 8;
  F4 ring[RING][2];
  int gitem = 0;
  auto gload = [&](int slot) __attribute__((always_inline)) {
    const int it = gitem < nitems ? gitem : nitems - 1;        // past the end: a redundant load, never multiplied
    const float* s = gsrc + (unsigned)it * (unsigned)kLoopItemFloats;
    ring[slot][0] = ld4(s);
    ring[slot][1] = ld4(s + 4);
    ++gitem;
  };
  auto mma_item = [&](int j, const F4 (&x)[RT][2], f32x4 (&acc)[RT]) __attribute__((always_inline)) {
    const int slot = j % RING;
    const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
    if (SB_EXP & 8) { acc[0][0] += ring[slot][0].x + ring[slot][1].w + x[0][0].x; gload(slot); sched_fence(); return; }
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][1]), wh, acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wl, acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wh, acc[t]);
    if (!(SB_EXP & 4)) gload(slot);
    sched_fence();
  };

  // ---- prologue: the strip's K segments -> split images; the first items of the stream are in flight meanwhile
#pragma unroll
  for (int j = 0; j < RING; ++j) gload(j);
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    const float* src = sg == 0 ? p.A : p.A2;
#pragma unroll
    for (int j = 0; j < RT * 2; ++j) {
      const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      const F4 v = (SB_EXP & 1) ? F4{0.01f * c4, 0.5f, -0.25f, 0.001f * row} : ld4_hint<NT>(src + (size_t)m * 256 + c4 * 4);
      unsigned h0, l0, h1, l1;
      split16_pair(v.x, v.y, h0, l0);
      split16_pair(v.z, v.w, h1, l1);
      unsigned* d = reinterpret_cast<unsigned*>(Xs + sg * BM * XS) + row * XS + (c4 >> 3) * 32 + (c4 & 7) * 2;
      *reinterpret_cast<U2*>(d) = U2{h0, h1};
      *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
    }
  }
  __syncthreads();

  const float* xa = Xs + r * XS + g * 4;
  for (int pr = 0; pr < npairs; ++pr) {
    f32x4 acc0[RT], acc1[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = acc0[t]; }
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        F4 x[RT][2];
#pragma unroll
        for (int t = 0; t < RT; ++t) { x[t][0] = ld4(xa + sg * BM * XS + t * 16 * XS + 32 * c); x[t][1] = ld4(xa + sg * BM * XS + t * 16 * XS + 32 * c + 16); }
        mma_item(2 * (sg * 8 + c), x, acc0);
        mma_item(2 * (sg * 8 + c) + 1, x, acc1);
      }
    const float bi0 = p.bias[pr * 256 + col0], bi1 = p.bias[pr * 256 + 128 + col0];
    if constexpr (!LN) {
      if constexpr (STAGE) {
        // one 128-column block at a time through the staging tile, out with 16-byte stores
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if (pr > 0 || cb > 0) __syncthreads();     // the previous block has left the tile
#pragma unroll
          for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) St[(t * 16 + g * 4 + i) * HS + col0] = (cb == 0 ? acc0[t][i] + bi0 : acc1[t][i] + bi1);
          __syncthreads();
#pragma unroll
          for (int j = 0; j < RT; ++j) {
            const int idx = tid + j * 512, row = idx >> 5, c4 = idx & 31;
            if (m0 + row < p.M && (!(SB_EXP & 2) || row == 0)) st4_hint<NT>(p.Y + (size_t)(m0 + row) * p.ldy + pr * 256 + cb * 128 + c4 * 4, ld4(St + row * HS + c4 * 4));
          }
        }
      } else {
        __syncthreads();                             // single pair: every wave is done with the A images
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* o = Xs + (t * 16 + g * 4 + i) * XS + col0;
            o[0] = acc0[t][i] + bi0;
            o[128] = acc1[t][i] + bi1;
          }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RT * 2; ++j) {
          const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
          if (m0 + row < p.M && (!(SB_EXP & 2) || row == 0)) st4_hint<NT>(p.Y + (size_t)(m0 + row) * p.ldy + c4 * 4, ld4(Xs + row * XS + c4 * 4));
        }
      }
    } else {
      // ---- out-projection epilogue: v = acc + bias + residual; LayerNorm(g1, b1); + cvec[sample]; LayerNorm(g2, b2)
      __syncthreads();                               // every wave is done with the A image: its rows now hold the residual rows
#pragma unroll
      for (int j = 0; j < RT * 2; ++j) {
        const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        st4(Xs + row * XS + c4 * 4, ld4(p.res + (size_t)m * 256 + c4 * 4));
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* q = Xs + (t * 16 + g * 4 + i) * XS + col0;
          acc0[t][i] += bi0 + q[0];
          acc1[t][i] += bi1 + q[128];
        }
      auto layer_norm = [&](const float* gamma, const float* beta) __attribute__((always_inline)) {
        const float ga = gamma[col0], gb = gamma[128 + col0], ba = beta[col0], bb = beta[128 + col0];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          F4 s;
          float* sp = &s.x;
#pragma unroll
          for (int i = 0; i < 4; ++i) sp[i] = sum16(acc0[t][i] + acc1[t][i]);
          if (r == 0) st4(red + wave * BM + t * 16 + g * 4, s);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          F4 m = ld4(red + t * 16 + g * 4);
#pragma unroll
          for (int w = 1; w < 8; ++w) m = f4add(m, ld4(red + w * BM + t * 16 + g * 4));
          const float mean[4] = {m.x * (1.0f / 256.0f), m.y * (1.0f / 256.0f), m.z * (1.0f / 256.0f), m.w * (1.0f / 256.0f)};
          F4 s;
          float* sp = &s.x;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc0[t][i] -= mean[i];
            acc1[t][i] -= mean[i];
            sp[i] = sum16(acc0[t][i] * acc0[t][i] + acc1[t][i] * acc1[t][i]);
          }
          if (r == 0) st4(red + 8 * BM + wave * BM + t * 16 + g * 4, s);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          F4 q = ld4(red + 8 * BM + t * 16 + g * 4);
#pragma unroll
          for (int w = 1; w < 8; ++w) q = f4add(q, ld4(red + 8 * BM + w * BM + t * 16 + g * 4));
          const float rs[4] = {rsqrtf(q.x * (1.0f / 256.0f) + kLnEps), rsqrtf(q.y * (1.0f / 256.0f) + kLnEps),
                               rsqrtf(q.z * (1.0f / 256.0f) + kLnEps), rsqrtf(q.w * (1.0f / 256.0f) + kLnEps)};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc0[t][i] = acc0[t][i] * rs[i] * ga + ba;
            acc1[t][i] = acc1[t][i] * rs[i] * gb + bb;
          }
        }
      };
      if (p.cvec && tid < BM) {                       // one division per row, not per element
        const int m = m0 + tid < p.M ? m0 + tid : p.M - 1;
        sidx[tid] = m / p.rpg;
      }
      layer_norm(p.g1, p.b1);                        // (its barriers publish sidx)
      if (p.cvec) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* cv = p.cvec + (size_t)sidx[t * 16 + g * 4 + i] * 256 + col0;
            acc0[t][i] += cv[0];
            acc1[t][i] += cv[128];
          }
        layer_norm(p.g2, p.b2);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* o = Xs + (t * 16 + g * 4 + i) * XS + col0;     // own positions: the residual there has been consumed by this lane
          o[0] = acc0[t][i];
          o[128] = acc1[t][i];
        }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RT * 2; ++j) {
        const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
        if (m0 + row < p.M) st4(p.Y + (size_t)(m0 + row) * p.ldy + c4 * 4, ld4(Xs + row * XS + c4 * 4));
      }
    }
  }
}

}  // namespace mld
