// Post-norm feed-forward block of a decoder / encoder layer, split-f16 operands:
//   Y = LayerNorm(X + linear2(gelu(linear1(X))))        (cross_attention.py:340-343 decoder layer, :268-271 encoder layer)
// for D = 256, FF = 1024, structured by what the sample-major loop taught
// (loop_fused.hpp): a weight element is used by exactly one wave (wave w owns columns 16w .. 16w + 15 of every 128-column block,
// ALL row tiles of the strip), so weights never go through LDS: every lane loads its two 16-byte MFMA operands per item
// straight from a fragment-ordered stream (`finalize` re-packs linear1 / linear2 per layer in consumption order) into a register
// ring -- no weight staging, no barrier per item (round 2's LDS-staged form, retired in round 4: 128 barriers and 2 MB of LDS stores +
// 4 MB of LDS reads per workgroup; 1.72 vs 1.10 ms per launch at 2 048 motions).  What is left in LDS is the strip itself -- RT x 16 rows of X as a split image -- and one 128-wide block of the
// hidden activation at a time.  RT = 6 (96 rows): 2 MB of weights per 96 rows instead of per 64, i.e. 1.5x less L2 -> CU traffic;
// RT = 3 (48 rows, 80 KB of LDS, 128 registers): two workgroups per CU, four waves per SIMD -- one workgroup's barriers and GELU
// stretches run under the other's matrix instructions; measured 2 % faster over the decoder than RT = 6.
//
// Pipeline over the eight hidden blocks:   run1(0); gelu(0)
//   hb = 1..7:  W(hb-1) | run1(hb) | { run2(hb-1) || gelu(hb) }        W = barrier, write H block to LDS, barrier
//   W(7) | run2(7) | bias + residual + LayerNorm -> Y
// run1 = linear1 of a block (8 items), run2 = linear2's share of a block (8 items: 4 K chunks x 2 column blocks, A fragments
// shared), gelu = bias + erf-GELU + hi/lo split of the block's accumulators into registers: pure VALU, issued between the
// matrix instructions of run2, which do not depend on it.  The H block is single-buffered: it is rewritten only between the
// two barriers of W, after every wave has left run2 of the previous block.
//
// Summation order differs from the two staged GEMMs of gemm.hpp (the "ffn_strip" = 0 path), so results agree to fp32 rounding, not bitwise.
#pragma once
#include "loop_fused.hpp"

namespace mld {

struct FfnArgs {
  const float* X = nullptr;        // [M][256] fp32: block input and residual
  const float* W1 = nullptr;       // split image of linear1.weight [1024][256]
  const float* b1 = nullptr;       // [1024]
  const float* W2 = nullptr;       // split image of linear2.weight [256][1024]
  const float* b2 = nullptr;       // [256]
  const float* gamma = nullptr;    // LayerNorm after the residual
  const float* beta = nullptr;
  float* Y = nullptr;              // [M][256]
  int M = 0;
  const int* skip_lens = nullptr;  // skip row tiles made only of rows (row % rpg) >= skip_lens[row / rpg] (padded frames)
  int skip_rpg = 1;
  // ffn_strip_x3_kernel<RT, true> ("decoder tail": the self-attention out-projection + residual + norm1 + cross-attention vector + norm2 in
  // front of the feed-forward block, one launch; X is not read, the block input is produced in LDS)
  const float* AO = nullptr;       // [M][256] attention output (A operand of the out-projection)
  const float* Wo = nullptr;       // fragment-ordered stream of out_proj.weight (16 items: 8 chunks x [block 0, block 1])
  const float* bo = nullptr;       // [256]
  const float* res = nullptr;      // [M][256] the layer input (residual of norm1)
  const float* g1 = nullptr; const float* be1 = nullptr;      // norm1
  const float* cvec = nullptr; int rpg = 1;                   // + cvec[row / rpg][256] before norm2
  const float* g2 = nullptr; const float* be2 = nullptr;      // norm2
};


#ifndef TB_EXP
#define TB_EXP 0          // tools/loopbench/tail_bench.hip experiments (measurement only, 0 in the library)
#endif
constexpr int kFsXs = 264, kFsHs = 136;      // row strides (words), = 8 mod 16 (conflict-free fragment reads)
template <int RT>
constexpr int ffn_strip_lds_bytes() { return (RT * 16 * kFsXs + RT * 16 * kFsHs + 2 * 8 * RT * 16 + RT * 16) * 4; }   // RT = 6: 160 128 B; RT = 4: 106 752 B; RT = 3: 80 064 B (two per CU)

// items of one layer's stream: run1(0), then [run1(hb), run2(hb - 1)] for hb = 1..7, then run2(7); 128 items of 16 KB
constexpr int kFfnStripItems = 128;

// grid = ceil(M / (16 RT)); block = 512.  p.W1 = the layer's fragment-ordered stream (W2 unused).
// TAIL: the rest of a decoder layer behind its self-attention in ONE launch -- out-projection (16 more items, from the row-strip GEMM's
// stream of out_proj.weight) + residual + norm1 + cross-attention vector + norm2 produce the block input in LDS instead of reading it:
// the layer's H1 tensor (M x 256 fp32, written by one launch and read by the next) disappears, 0.8 GB of the decoder's 5.3 GB of HBM
// traffic per layer at 2 048 motions -- and the decoder's row-strip kernels are bound by exactly that traffic (~3.1 TB/s, r03).
// SWZ: the strip's and the hidden block's images are stored XOR-swizzled by the row, exactly as in the persistent loop (loop_fused.hpp
// SWZ: physical word = logical word ^ 4 ((row >> 2) & 3); 8-byte row stores 4-way -> 2-way, fragment reads stay conflict free).
template <int RT, bool TAIL = false, bool SWZ = false>
__global__ __launch_bounds__(512, RT <= 3 ? 4 : 2) void ffn_strip_x3_kernel(FfnArgs p) {
  constexpr int BM = RT * 16, XS = kFsXs, HS = kFsHs;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem_fs[];
  float* smem = smem_fs;
#endif
  float* Xs = smem;                    // [BM][264] split image of the strip (A operand of linear1, residual)
  float* Hs = Xs + BM * XS;            // [BM][136] split image of one hidden block (A operand of linear2)
  float* red = Hs + BM * HS;           // [2][8][BM] LayerNorm partial sums
  int* sidx = reinterpret_cast<int*>(red + 2 * 8 * BM);     // [BM] sample of each row (TAIL: which cvec row to add)
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int col0 = wave * 16 + r;

  if (p.skip_lens) {                   // uniform exit for strips of padded frames only (gemm.hpp)
    const int t0 = m0, t1 = (t0 + BM < p.M ? t0 + BM : p.M) - 1;
    bool all_padding = true;
    for (int b = t0 / p.skip_rpg; b <= t1 / p.skip_rpg; ++b) {
      const int first = (t0 > b * p.skip_rpg ? t0 : b * p.skip_rpg) - b * p.skip_rpg;
      if (first < p.skip_lens[b]) { all_padding = false; break; }
    }
    if (all_padding) return;
  }

  // ---- weight ring (loop_fused.hpp): 4 items in flight per lane
  constexpr int RING = RT <= 3 ? 2 : 4;      // (must divide 8: items are numbered per run of 8)
  const float* gsrc = p.W1 + tid * 8;
  F4 ring[RING][2];
  int gitem = 0;
  constexpr int kItems = kFfnStripItems + (TAIL ? 16 : 0);
  auto gload = [&](int slot) __attribute__((always_inline)) {
    const int it = gitem < kItems ? gitem : kItems - 1;     // past the end: a redundant load, never multiplied
    const float* s = (TAIL && it < 16) ? p.Wo + tid * 8 + (unsigned)it * (unsigned)kLoopItemFloats
                                       : gsrc + (unsigned)(it - (TAIL ? 16 : 0)) * (unsigned)kLoopItemFloats;
    ring[slot][0] = ld4(s);
    ring[slot][1] = ld4(s + 4);
    ++gitem;
  };
  // tr: the transposed product (loop_fused.hpp mma_item): lane (r, g) holds row r, columns 4g .. 4g + 3 of the wave's 16 -- linear1,
  // whose outputs go through GELU into the hidden image as 8-byte row stores instead of 2-byte ones that collide on the banks
  auto mma_item = [&](int j, const F4 (&x)[RT][2], f32x4 (&acc)[RT], bool tr = false) __attribute__((always_inline)) {
    const int slot = j % RING;
    const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
    if (tr) {
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][1]), acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(wl, __builtin_bit_cast(U4, x[t][0]), acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][0]), acc[t]);
    } else {
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][1]), wh, acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wl, acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wh, acc[t]);
    }
    if (!(TB_EXP & 2)) gload(slot);
    sched_fence();                     // keeps the ring's loads where they are written (rt.hpp)
  };
  auto frags = [&](const float* a0, int st, int c, F4 (&x)[RT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < RT; ++t) { x[t][0] = ld4(a0 + t * 16 * st + 32 * c); x[t][1] = ld4(a0 + t * 16 * st + 32 * c + 16); }
  };

  // ---- prologue: the strip (TAIL: of the attention output) -> split image; the first items of the stream are in flight meanwhile
#pragma unroll
  for (int j = 0; j < RING; ++j) gload(j);
#pragma unroll
  for (int j = 0; j < RT * 2; ++j) {
    const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    const F4 v = (TB_EXP & 4) ? F4{0.1f * c4, 0.2f, -0.3f, 0.01f * row} : ld4((TAIL ? p.AO : p.X) + (size_t)m * 256 + c4 * 4);
    unsigned h0, l0, h1, l1;
    split16_pair(v.x, v.y, h0, l0);
    split16_pair(v.z, v.w, h1, l1);
    unsigned* d = reinterpret_cast<unsigned*>(Xs) + row * XS + (c4 >> 3) * 32 + (c4 & 7) * 2;
    if constexpr (SWZ) d = reinterpret_cast<unsigned*>(Xs) + row * XS + (((c4 >> 3) * 32 + (c4 & 7) * 2) ^ (((row >> 2) & 3) << 2));
    *reinterpret_cast<U2*>(d) = U2{h0, h1};
    *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
  }
  if constexpr (TAIL) {
    if (p.cvec && tid < BM) {          // one division per row, not per element
      const int m = m0 + tid < p.M ? m0 + tid : p.M - 1;
      sidx[tid] = m / p.rpg;
    }
  }
  __syncthreads();

  // SWZ: this lane's 16-byte group of a half chunk is g ^ (r >> 2) in rows 16 t + r
  const float* xa = SWZ ? Xs + r * XS + ((g ^ (r >> 2)) << 2) : Xs + r * XS + g * 4;
  const float* ha = SWZ ? Hs + r * HS + ((g ^ (r >> 2)) << 2) : Hs + r * HS + g * 4;
  if constexpr (TAIL) {
    // ---- out-projection (transposed products: lane (r, g) holds row r, columns 16 wave + 4g .. + 3 of each 128-column block) + bias +
    //      residual, LayerNorm(g1), + cvec[sample], LayerNorm(g2) (gemm_strip_x3.hpp's LN form), result -> the strip's image in Xs
    const int cq0 = wave * 16 + g * 4;
    f32x4 o0[RT], o1[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { o0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; o1[t] = o0[t]; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      F4 x[RT][2];
      frags(xa, XS, c, x);
      mma_item(2 * c, x, o0, true);
      mma_item(2 * c + 1, x, o1, true);
    }
    {
      const F4 ba = ld4(p.bo + cq0), bb = ld4(p.bo + 128 + cq0);
      const float bav[4] = {ba.x, ba.y, ba.z, ba.w}, bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        int m = m0 + t * 16 + r;
        m = m < p.M ? m : p.M - 1;
        const F4 ra = (TB_EXP & 4) ? F4{0.1f, 0.2f, 0.3f, 0.4f} : ld4(p.res + (size_t)m * 256 + cq0), rb = (TB_EXP & 4) ? F4{0.f, 1.f, 0.f, 1.f} : ld4(p.res + (size_t)m * 256 + 128 + cq0);
        const float rav[4] = {ra.x, ra.y, ra.z, ra.w}, rbv[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o0[t][i] += bav[i] + rav[i]; o1[t][i] += bbv[i] + rbv[i]; }
      }
    }
    auto layer_norm = [&](const float* gamma, const float* beta) __attribute__((always_inline)) {
      const F4 ga = ld4(gamma + cq0), gb = ld4(gamma + 128 + cq0), ba = ld4(beta + cq0), bb = ld4(beta + 128 + cq0);
      const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, gbv[4] = {gb.x, gb.y, gb.z, gb.w}, bav[4] = {ba.x, ba.y, ba.z, ba.w}, bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        float sum = ((o0[t][0] + o0[t][1]) + (o0[t][2] + o0[t][3])) + ((o1[t][0] + o1[t][1]) + (o1[t][2] + o1[t][3]));
        sum = sum_groups(sum);
        if (g == 0) red[(t * 16 + r) * 8 + wave] = sum;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const F4 ma = ld4(red + (t * 16 + r) * 8), mb = ld4(red + (t * 16 + r) * 8 + 4);
        const float mean = (((ma.x + ma.y) + (ma.z + ma.w)) + ((mb.x + mb.y) + (mb.z + mb.w))) * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o0[t][i] -= mean;
          o1[t][i] -= mean;
          sq += o0[t][i] * o0[t][i] + o1[t][i] * o1[t][i];
        }
        sq = sum_groups(sq);
        if (g == 0) red[8 * BM + (t * 16 + r) * 8 + wave] = sq;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const F4 qa = ld4(red + 8 * BM + (t * 16 + r) * 8), qb = ld4(red + 8 * BM + (t * 16 + r) * 8 + 4);
        const float rs = rsqrtf((((qa.x + qa.y) + (qa.z + qa.w)) + ((qb.x + qb.y) + (qb.z + qb.w))) * (1.0f / 256.0f) + kLnEps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o0[t][i] = o0[t][i] * rs * gav[i] + bav[i];
          o1[t][i] = o1[t][i] * rs * gbv[i] + bbv[i];
        }
      }
    };
    layer_norm(p.g1, p.be1);
    if (p.cvec) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float* cv = p.cvec + (size_t)sidx[t * 16 + r] * 256 + cq0;
        const F4 ca = ld4(cv), cb = ld4(cv + 128);
        o0[t][0] += ca.x; o0[t][1] += ca.y; o0[t][2] += ca.z; o0[t][3] += ca.w;
        o1[t][0] += cb.x; o1[t][1] += cb.y; o1[t][2] += cb.z; o1[t][3] += cb.w;
      }
      __syncthreads();                 // norm1's second pass has been read by every wave before `red` is rewritten
      layer_norm(p.g2, p.be2);
    }
    // every wave left the out-projection before norm1's first barrier: the attention-output image is dead; the block input takes its place
    const int rw0 = SWZ ? (((wave >> 1) * 32 + (wave & 1) * 8 + g * 2) ^ ((r >> 2) << 2)) : (wave >> 1) * 32 + (wave & 1) * 8 + g * 2;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      unsigned h0, l0, h1, l1;
      unsigned* w = reinterpret_cast<unsigned*>(Xs + (t * 16 + r) * XS) + rw0;
      split16_two(o0[t][0], o0[t][1], h0, l0);
      split16_two(o0[t][2], o0[t][3], h1, l1);
      *reinterpret_cast<U2*>(w) = U2{h0, h1};
      *reinterpret_cast<U2*>(w + 16) = U2{l0, l1};
      split16_two(o1[t][0], o1[t][1], h0, l0);
      split16_two(o1[t][2], o1[t][3], h1, l1);
      *reinterpret_cast<U2*>(w + 128) = U2{h0, h1};
      *reinterpret_cast<U2*>(w + 144) = U2{l0, l1};
    }
    __syncthreads();
  }
  // half-word offset of column col0 in a row image; the rows read with it are 16 t + 4 g + i (plain accumulator layout): swizzled by g
  const int hw0 = SWZ ? ((((wave >> 1) * 32 + (wave & 1) * 8 + (r >> 1)) ^ (g << 2)) * 2 + (r & 1)) : ((wave >> 1) * 32 + (wave & 1) * 8 + (r >> 1)) * 2 + (r & 1);

  f32x4 h[RT], y0[RT], y1[RT];
  unsigned hvh[RT][2], hvl[RT][2];     // one hidden block after bias + GELU: high / low halves of elements (i, i + 1) packed per word
#pragma unroll
  for (int t = 0; t < RT; ++t) { y0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; y1[t] = y0[t]; }
  auto run1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < RT; ++t) h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // RT <= 4: the fragments of chunk c + 1 are requested before chunk c is multiplied (the fence in mma_item keeps that order);
    // RT = 6 has no registers for a second fragment set: the SIMD's other wave covers the LDS latency
    constexpr int NB = RT == 4 ? 2 : 1;
    F4 x[NB][RT][2];
    if constexpr (NB == 2) frags(xa, XS, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if constexpr (NB == 2) { if (c + 1 < 8) frags(xa, XS, c + 1, x[(c + 1) & 1]); }
      else if (!(TB_EXP & 16) || c == 0) frags(xa, XS, c, x[0]);
      mma_item(c, x[c & (NB - 1)], h, true);
    }
  };
  constexpr int NB = RT == 4 ? 2 : 1;
  // elements 2 ip, 2 ip + 1 of tile t: row 16 t + r, hidden columns 16 wave + 4 g + 2 ip, + 1 (transposed linear1 accumulators)
  auto gelu_two = [&](int t, int ip, F4 b1) __attribute__((always_inline)) {
    if (TB_EXP & 1) split16_two(h[t][2 * ip] + (ip ? b1.z : b1.x), h[t][2 * ip + 1] + (ip ? b1.w : b1.y), hvh[t][ip], hvl[t][ip]);
    else split16_two(gelu_erf(h[t][2 * ip] + (ip ? b1.z : b1.x)), gelu_erf(h[t][2 * ip + 1] + (ip ? b1.w : b1.y)), hvh[t][ip], hvl[t][ip]);
  };
  auto write_block = [&]() __attribute__((always_inline)) {
    if (!(TB_EXP & 8)) __syncthreads();                   // every wave has left run2 of the previous block
#pragma unroll
    for (int t = 0; t < RT; ++t)
    {
      // four consecutive columns of row 16 t + r: words (wave >> 1) 32 + (wave & 1) 8 + 2 g, + 1 of the high plane, + 16 for the low one
      unsigned* w = reinterpret_cast<unsigned*>(Hs + (t * 16 + r) * HS) + (wave >> 1) * 32 + (wave & 1) * 8 + g * 2;
      if constexpr (SWZ) w = reinterpret_cast<unsigned*>(Hs + (t * 16 + r) * HS) + (((wave >> 1) * 32 + (wave & 1) * 8 + g * 2) ^ ((r >> 2) << 2));
      *reinterpret_cast<U2*>(w) = U2{hvh[t][0], hvh[t][1]};
      *reinterpret_cast<U2*>(w + 16) = U2{hvl[t][0], hvl[t][1]};
    }
    if (!(TB_EXP & 8)) __syncthreads();
  };

  run1();
  {
    const F4 b1 = ld4(p.b1 + wave * 16 + g * 4);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) gelu_two(t, ip, b1);
  }
  for (int hb = 1; hb < 8; ++hb) {
    write_block();                     // block hb - 1 -> LDS
    const F4 b1 = ld4(p.b1 + hb * 128 + wave * 16 + g * 4);
    run1();                            // linear1 of block hb
    // linear2's share of block hb - 1 (8 items: 4 chunks x 2 column blocks) with the GELU of block hb spread between its items
    constexpr int PER = (RT * 2 + 7) / 8;     // element PAIRS per item
    F4 x[NB][RT][2];
    if constexpr (NB == 2) frags(ha, HS, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (NB == 2) { if (c + 1 < 4) frags(ha, HS, c + 1, x[(c + 1) & 1]); }
      else if (!(TB_EXP & 32) || c == 0) frags(ha, HS, c, x[0]);
      mma_item(2 * c, x[c & (NB - 1)], y0);
#pragma unroll
      for (int e = (2 * c) * PER; e < (2 * c + 1) * PER; ++e)
        if (e < RT * 2) gelu_two(e >> 1, e & 1, b1);
      mma_item(2 * c + 1, x[c & (NB - 1)], y1);
#pragma unroll
      for (int e = (2 * c + 1) * PER; e < (2 * c + 2) * PER; ++e)
        if (e < RT * 2) gelu_two(e >> 1, e & 1, b1);
    }
  }
  write_block();
  {
    F4 x[NB][RT][2];
    if constexpr (NB == 2) frags(ha, HS, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (NB == 2) { if (c + 1 < 4) frags(ha, HS, c + 1, x[(c + 1) & 1]); }
      else frags(ha, HS, c, x[0]);
      mma_item(2 * c, x[c & (NB - 1)], y0);
      mma_item(2 * c + 1, x[c & (NB - 1)], y1);
    }
  }

  // ---- bias + residual (the strip's own image: high + low half) + LayerNorm over the 256 columns, 8 waves x 2 column blocks
  const float lb0 = p.b2[col0], lb1 = p.b2[128 + col0];
  const float g0 = p.gamma[col0], g1 = p.gamma[128 + col0], e0 = p.beta[col0], e1 = p.beta[128 + col0];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    F4 s;
    float* sp = &s.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned short* q = reinterpret_cast<const unsigned short*>(Xs + (t * 16 + g * 4 + i) * XS) + hw0;
      y0[t][i] += lb0 + f16_bits_value(q[0]) + f16_bits_value(q[32]);
      y1[t][i] += lb1 + f16_bits_value(q[256]) + f16_bits_value(q[288]);      // + 128 words: the second column block
      sp[i] = sum16(y0[t][i] + y1[t][i]);
    }
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
      y0[t][i] -= mean[i];
      y1[t][i] -= mean[i];
      sp[i] = sum16(y0[t][i] * y0[t][i] + y1[t][i] * y1[t][i]);
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
      // the normalised rows are parked in the strip's own LDS rows (every residual read of it happened before the two barriers
      // above), fp32, row stride 264 words, and leave with 16-byte stores: a wave instruction then covers 1 KiB of contiguous output
      // instead of four 64-byte fragments (gemm.hpp store_tile_from_lds: scattered 4-byte stores cost more than the main loop)
      float* o = Xs + (t * 16 + g * 4 + i) * XS + col0;
      o[0] = y0[t][i] * rs[i] * g0 + e0;
      o[128] = y1[t][i] * rs[i] * g1 + e1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RT * 2; ++j) {
    const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
    if (m0 + row < p.M && (!(TB_EXP & 4) || row == 0)) st4(p.Y + (size_t)(m0 + row) * 256 + c4 * 4, ld4(Xs + row * XS + c4 * 4));
  }
}

}  // namespace mld
