// Y = epilogue(X · Wᵀ + b): the one GEMM family of the MLD engine (exact-fp32 MFMA path).
//
// Replaces every nn.Linear / packed in-proj / out-proj of the reference's hot path
// (cross_attention.py:56-58,113-115,259-272,323-345; mld_denoiser.py:65-68; embeddings.py:298-305;
//  mld_vae.py:243) and fuses what follows it there: bias, erf-GELU / SiLU, residual add,
// post-norm LayerNorm, the 1-key cross-attention shortcut (+cvec, second LayerNorm), the
// `output[~mask.T] = 0` of mld_vae.py:245 and the `cat([x, skip])` of the skip connection
// (as a second K segment, no concat buffer).
//
// Data layout: activations [rows, K] row-major fp32; weights exactly as nn.Linear stores them,
// [N, K] row-major -- both operands are K-contiguous, so an MFMA fragment is two 16-byte loads.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32): within a 32-wide K chunk lane l=(g=l>>4, r=l&15) holds the
// 8 consecutive K values k0+8g..k0+8g+7 of row r for A and of weight row r for B; MFMA number i of
// the chunk consumes element i of both fragments.  Any pairing of k-slots is legal as long as A and B
// agree, and this one turns the fragment fetch into contiguous 32-byte reads.
//
// Roofline: MFMA-bound (fp32 matrix rate 157 TF); operands are L2/MALL resident (weights 74 MB).
#pragma once
#include "rt.hpp"

namespace mld {

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };

struct GemmArgs {
  const float* A = nullptr;  int lda = 0;  int K1 = 0;   // first K segment  [M, K1]
  const float* A2 = nullptr; int lda2 = 0; int K2 = 0;   // optional second K segment (skip concat)
  const float* W = nullptr;  int ldw = 0;                // [N, K1+K2]
  const float* bias = nullptr;                           // [N]
  float* Y = nullptr; int ldy = 0;
  int M = 0, N = 0;
  int relu_in = 0;                                       // ReLU on A while loading (mld_denoiser.py:65-68)
  int act = ACT_NONE;
  const int* lens = nullptr; int rows_per_group = 1;     // zero rows with (row % rpg) >= lens[row / rpg]
  const int* skip_lens = nullptr; int skip_rpg = 1;      // skip row tiles made only of rows (row % rpg) >= skip_lens[row / rpg]
  long long sA = 0, sW = 0, sBias = 0, sY = 0;           // blockIdx.z strides (elements)
  // LayerNorm epilogue (N must equal the block's BN)
  const float* res = nullptr; int ldres = 0;
  const float* g1 = nullptr; const float* b1 = nullptr;
  const float* cvec = nullptr; int ldcvec = 0;           // + cvec[row / rows_per_group] then LN(g2,b2)
  const float* g2 = nullptr; const float* b2 = nullptr;
  unsigned long long* trace = nullptr;   // measurement only (staged kernels): 8 timestamps per wave
  int w_split = 0;                       // PREC_BF16X3: W points into the pre-split copy of the weight arena (elementwise.hpp split_bf16_weights_kernel)
};

struct Frag { float v[8]; };

// Copy a [BM][BN] fp32 tile parked in LDS (row stride BN + 4) to Y with 16-byte stores: one wave instruction
// covers 1 KiB of contiguous output.  (Measured: 32 scattered 4-byte stores per lane -- 64-byte row fragments --
// cost 13-26 k cycles per workgroup, more than the whole fp32 MFMA main loop; profiles/r01_v8_gemm_phase_trace.)
template <int BM, int BN, int NT>
__device__ __forceinline__ void store_tile_from_lds(const float* Cs, float* Y, int ldy, int row0, int col0, int M, int N, int tid) {
  constexpr int CST = BN + 4, V = BN / 4;
  const bool vec_ok = (ldy & 3) == 0 && col0 + BN <= N && ((reinterpret_cast<unsigned long long>(Y) & 15) == 0);
  if (vec_ok) {
#pragma unroll
    for (int j = 0; j < BM * V / NT; ++j) {
      const int idx = tid + j * NT, row = idx / V, c4 = idx - row * V;
      if (row0 + row < M) st4(Y + (long long)(row0 + row) * ldy + col0 + c4 * 4, ld4(Cs + row * CST + c4 * 4));
    }
  } else {   // ragged right edge / unaligned row stride (final_layer: N = ldy = 263): scalar, still row-contiguous
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int row = idx / BN, c = idx - row * BN;
      if (row0 + row < M && col0 + c < N) Y[(long long)(row0 + row) * ldy + col0 + c] = Cs[row * CST + c];
    }
  }
}
struct alignas(8) uint2_t { unsigned x, y; };

template <int REP>
__device__ __forceinline__ void load_frags(Frag (&f)[REP], const float* base, int ld, int row0, int nrows,
                                           int k, int r, int g, bool relu) {
#pragma unroll
  for (int t = 0; t < REP; ++t) {
    int row = row0 + t * 16 + r;
    row = row < nrows ? row : nrows - 1;
    const float* p = base + (long long)row * ld + k + g * 8;
    F4 a = ld4(p), b = ld4(p + 4);
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
    }
    f[t].v[0] = a.x; f[t].v[1] = a.y; f[t].v[2] = a.z; f[t].v[3] = a.w;
    f[t].v[4] = b.x; f[t].v[5] = b.y; f[t].v[6] = b.z; f[t].v[7] = b.w;
  }
}

// floats per staged row: one 32-wide K chunk + 8 pad.  The packed operand images (bf16 hi | lo) are read as ds_read_b128 at word
// 4g of row r: that instruction is served in four 16-lane groups over 64 banks (MI355X_MICROARCH.md, LDS), and with a row stride
// of 36 words rows r and r + 9 (mod 16) of one group land on the same banks -- 2-way conflicts, half the LDS rate, on kernels
// whose split-bf16 MFMAs (16 cycles) make them LDS bound.  Strides = 8 mod 16 words are conflict free for that pattern.  The fp32
// fragments used to be two ds_read_b128 at words 8g and 8g + 4 -- 2-way at EVERY stride; they are now read at words 4g and 16 + 4g
// (lfrags below, strip.hpp, tile32.hpp), the same conflict-free pattern.
constexpr int kGemmLdsStride = 40;
template <int WM, int WN, int MREP, int NREP>
constexpr int gemm_lds_bytes() {    // the chunk double buffer, or the output tile parked for the 16-byte stores, whichever is larger
  constexpr int loop = 2 * (WM * MREP * 16 + WN * NREP * 16) * kGemmLdsStride * 4, epi = WM * MREP * 16 * (WN * NREP * 16 + 4) * 4;
  return loop > epi ? loop : epi;
}

// WM x WN waves per workgroup, each wave owns MREP x NREP tiles of 16x16.
// STAGED = false: fragments are fetched straight from global memory (first version; kept for the tiny
//                 one-off GEMMs: time MLP, text projection, per-sample cross-attention vectors).
// STAGED = true : the workgroup streams 32-wide K chunks of its A and W panels through LDS in full
//                 128-byte lines (coalesced 16 B/lane), double buffered: chunk k+1's global loads are in
//                 flight while chunk k's MFMAs run; one barrier per chunk.  (The register-direct form
//                 reached only ~50 of 157 TF on the decoder: fragment-shaped loads saturate the TA path,
//                 cdna_hip_programming.md "x through LDS in full lines".)
// PREC (staged path only): PREC_F32 = exact fp32 MFMA; PREC_BF16X3 = split-bf16 (rt.hpp): operands are split into
//                 bf16 hi/lo planes while they are written to LDS (same LDS footprint as fp32), 3 bf16 MFMAs
//                 per tile per K chunk; PREC_BF16 = operands rounded to bf16 (RNE) at the LDS store, one
//                 v_mfma_f32_16x16x32_bf16 per tile and chunk.  Accumulation, bias,
//                 residual, LayerNorm and the stored result are fp32 in every mode.
// KCS (staged path only): K / 32, compile time, so that the whole chunk pipeline is straight-line code: any
//                 runtime branch around a prefetch load makes hipcc's vmcnt bookkeeping conservative and the
//                 ring drains at every LDS store (seen in the ISA as vmcnt(5)..vmcnt(0) ladders).
template <int WM, int WN, int MREP, int NREP, bool LN, bool STAGED = false, int PREC = 0, int KCS = 0, bool TRACE = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmArgs p) {
  static_assert(PREC == PREC_F32 || STAGED, "reduced-precision operands need the LDS-staged main loop");
  constexpr int BM = WM * MREP * 16, BN = WN * NREP * 16;
  constexpr bool SPLIT = (MREP * NREP == 1);   // one tile per wave: split the k-chain over 2 accumulators
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM + wm * MREP * 16;
  const int n0 = blockIdx.y * BN + wn * NREP * 16;
  const long long z = blockIdx.z;
  const float* A = p.A + z * p.sA;
  const float* W = p.W + z * p.sW;
  const float* bias = p.bias ? p.bias + z * p.sBias : nullptr;
  float* Y = p.Y + z * p.sY;
  const int K = p.K1 + p.K2;
  const int KC = K / 32;
  const bool relu = p.relu_in != 0;

  if (p.skip_lens) {
    // Padded frames of a ragged batch: their rows never reach a valid row (attention masks them as keys, everything else
    // is row-wise) and the final layer zeroes them, so a tile made only of such rows is not computed at all.  Uniform exit.
    const int t0 = blockIdx.x * BM, t1 = (t0 + BM < p.M ? t0 + BM : p.M) - 1;
    bool all_padding = true;
    for (int b = t0 / p.skip_rpg; b <= t1 / p.skip_rpg; ++b) {
      const int first = (t0 > b * p.skip_rpg ? t0 : b * p.skip_rpg) - b * p.skip_rpg;
      if (first < p.skip_lens[b]) { all_padding = false; break; }
    }
    if (all_padding) return;
  }
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, rt0 = 0;
  if constexpr (TRACE) { rt0 = realtime_100mhz(); ts[0] = clock_pinned(); }
  f32x4 acc[MREP][NREP];
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int b = 0; b < NREP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  Frag fa0[MREP], fb0[NREP], fa1[MREP], fb1[NREP];

  auto load = [&](Frag(&fa)[MREP], Frag(&fb)[NREP], int kc) {
    int k = kc * 32;
    if (k < p.K1) load_frags<MREP>(fa, A, p.lda, m0, p.M, k, r, g, relu);
    else load_frags<MREP>(fa, p.A2, p.lda2, m0, p.M, k - p.K1, r, g, relu);
    load_frags<NREP>(fb, W, p.ldw, n0, p.N, k, r, g, false);
  };
  auto compute = [&](Frag(&fa)[MREP], Frag(&fb)[NREP]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int a = 0; a < MREP; ++a)
#pragma unroll
        for (int b = 0; b < NREP; ++b) {
          if (SPLIT && (i & 1)) acc2 = mfma_f32_16x16x4(fa[a].v[i], fb[b].v[i], acc2);
          else acc[a][b] = mfma_f32_16x16x4(fa[a].v[i], fb[b].v[i], acc[a][b]);
        }
  };

  if constexpr (!STAGED) {
    load(fa0, fb0, 0);
    for (int kc = 0; kc < KC; kc += 2) {
      if (kc + 1 < KC) load(fa1, fb1, kc + 1);
      compute(fa0, fb0);
      if (kc + 1 < KC) {
        if (kc + 2 < KC) load(fa0, fb0, kc + 2);
        compute(fa1, fb1);
      }
    }
  } else {
#if defined(MLDHIP_SIM)
    float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
    extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
    constexpr int NT = WM * WN * 64, ROWS = BM + BN, NLD = ROWS * 8 / NT;
    static_assert(ROWS * 8 % NT == 0, "panel rows must tile the workgroup");
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    // 4-deep register prefetch ring: chunks k+1..k+4 are in flight while chunk k is multiplied.  With one chunk
    // of look-ahead only ~48 KB per CU was outstanding against a latency x bandwidth product of ~250 KB
    // (1.5-2 us x ~56 B/clk/CU), which left the first staged version latency bound (profiles/r01_v7).
    F4 st0[NLD], st1[NLD], st2[NLD], st3[NLD];
    constexpr int NLA = BM * 8 / NT;          // staging slots that hold A rows (uniform across the workgroup)
    static_assert(BM * 8 % NT == 0, "A panel must fill whole staging slots");
    const float relu_lo = relu ? 0.f : -INFINITY;   // fmaxf(v, -inf) == v: ReLU-on-load without a branch
    auto gload = [&](F4(&stage)[NLD], int kc) {
      // Straight-line: no per-load condition (a `cond ? load : other` makes hipcc branch and drain vmcnt
      // per element, which serialised this prefetch in front of the MFMAs in the first version).
      const int k = kc * 32;
      const bool first = k < p.K1;
      const float* abase = first ? A : p.A2;
      const int ald = first ? p.lda : p.lda2;
      const int ak = first ? k : k - p.K1;
#pragma unroll
      for (int j = 0; j < NLA; ++j) {
        const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
        int m = bm0 + row;
        m = m < p.M ? m : p.M - 1;
        stage[j] = ld4(abase + (long long)m * ald + ak + c4 * 4);   // ReLU is applied in lstore: no use of the
      }                                                               // loaded value before the MFMAs of this chunk
#pragma unroll
      for (int j = NLA; j < NLD; ++j) {
        const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
        int n = bn0 + row - BM;
        n = n < p.N ? n : p.N - 1;
        stage[j] = ld4(W + (long long)n * p.ldw + k + c4 * 4);
      }
    };
    auto lstore = [&](int buf, F4(&stage)[NLD]) {
      float* dst = smem + buf * ROWS * kGemmLdsStride;
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
        F4 v = stage[j];
        if (j < NLA) { v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo); }
        if constexpr (PREC == PREC_F32) {
          st4(dst + row * kGemmLdsStride + c4 * 4, v);
        } else if constexpr (PREC == PREC_BF16) {
          // row image: [32 x bf16 | unused]; this thread owns k = 4*c4 .. 4*c4+3 -> words 2*c4, 2*c4+1
          unsigned* rowp = reinterpret_cast<unsigned*>(dst + row * kGemmLdsStride);
          *reinterpret_cast<uint2_t*>(rowp + c4 * 2) = uint2_t{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        } else {
          // row image: [32 x bf16 hi | 32 x bf16 lo | pad]; this thread owns k = 4*c4 .. 4*c4+3
          if (p.w_split && j >= NLA) { st4(dst + row * kGemmLdsStride + c4 * 4, v); continue; }   // W was split at finalize (split_bf16_weights_kernel): already the row image
          unsigned h0, l0, h1, l1;
          split16_pair(v.x, v.y, h0, l0);
          split16_pair(v.z, v.w, h1, l1);
          unsigned* rowp = reinterpret_cast<unsigned*>(dst + row * kGemmLdsStride);
          *reinterpret_cast<uint2_t*>(rowp + c4 * 2) = uint2_t{h0, h1};
          *reinterpret_cast<uint2_t*>(rowp + 16 + c4 * 2) = uint2_t{l0, l1};
        }
      }
    };
    auto lfrags = [&](int buf) {
      // fp32 fragments: words 4g .. 4g + 3 and 16 + 4g .. 16 + 4g + 3 of the chunk (conflict-free ds_read_b128, see kGemmLdsStride;
      // the k-slot pairing is free as long as A and W agree)
      const float* as = smem + buf * ROWS * kGemmLdsStride + (wm * MREP * 16 + r) * kGemmLdsStride + g * 4;
      const float* ws = smem + buf * ROWS * kGemmLdsStride + (BM + wn * NREP * 16 + r) * kGemmLdsStride + g * 4;
#pragma unroll
      for (int t = 0; t < MREP; ++t) {
        const F4 a = ld4(as + t * 16 * kGemmLdsStride), b = ld4(as + t * 16 * kGemmLdsStride + 16);
        fa0[t].v[0] = a.x; fa0[t].v[1] = a.y; fa0[t].v[2] = a.z; fa0[t].v[3] = a.w;
        fa0[t].v[4] = b.x; fa0[t].v[5] = b.y; fa0[t].v[6] = b.z; fa0[t].v[7] = b.w;
      }
#pragma unroll
      for (int t = 0; t < NREP; ++t) {
        const F4 a = ld4(ws + t * 16 * kGemmLdsStride), b = ld4(ws + t * 16 * kGemmLdsStride + 16);
        fb0[t].v[0] = a.x; fb0[t].v[1] = a.y; fb0[t].v[2] = a.z; fb0[t].v[3] = a.w;
        fb0[t].v[4] = b.x; fb0[t].v[5] = b.y; fb0[t].v[6] = b.z; fb0[t].v[7] = b.w;
      }
    };
    U4 ahi[MREP], alo[MREP], bhi[NREP], blo[NREP];
    auto lfrags_bf16 = [&](int buf) {
      const float* as = smem + buf * ROWS * kGemmLdsStride + (wm * MREP * 16 + r) * kGemmLdsStride;
      const float* ws = smem + buf * ROWS * kGemmLdsStride + (BM + wn * NREP * 16 + r) * kGemmLdsStride;
#pragma unroll
      for (int t = 0; t < MREP; ++t) {
        const U4* rp = reinterpret_cast<const U4*>(as + t * 16 * kGemmLdsStride);
        ahi[t] = rp[g];
        alo[t] = rp[4 + g];
      }
#pragma unroll
      for (int t = 0; t < NREP; ++t) {
        const U4* rp = reinterpret_cast<const U4*>(ws + t * 16 * kGemmLdsStride);
        bhi[t] = rp[g];
        blo[t] = rp[4 + g];
      }
    };
    auto mma_narrow = [&](int buf) {       // PREC_BF16: one matrix instruction per tile and K chunk
      const float* as = smem + buf * ROWS * kGemmLdsStride + (wm * MREP * 16 + r) * kGemmLdsStride;
      const float* ws = smem + buf * ROWS * kGemmLdsStride + (BM + wn * NREP * 16 + r) * kGemmLdsStride;
#pragma unroll
      for (int t = 0; t < MREP; ++t) ahi[t] = reinterpret_cast<const U4*>(as + t * 16 * kGemmLdsStride)[g];
#pragma unroll
      for (int t = 0; t < NREP; ++t) bhi[t] = reinterpret_cast<const U4*>(ws + t * 16 * kGemmLdsStride)[g];
#pragma unroll
      for (int a = 0; a < MREP; ++a)
#pragma unroll
        for (int b = 0; b < NREP; ++b) acc[a][b] = mfma_bf16_16x16x32(ahi[a], bhi[b], acc[a][b]);
    };
    auto compute_bf16 = [&]() {
#pragma unroll
      for (int a = 0; a < MREP; ++a)
#pragma unroll
        for (int b = 0; b < NREP; ++b) {
          acc[a][b] = mfma_x3_16x16x32(alo[a], bhi[b], acc[a][b]);
          acc[a][b] = mfma_x3_16x16x32(ahi[a], blo[b], acc[a][b]);
          acc[a][b] = mfma_x3_16x16x32(ahi[a], bhi[b], acc[a][b]);
        }
    };
    auto mma = [&](int buf) {
      if constexpr (PREC == PREC_F32) {
        lfrags(buf);
        compute(fa0, fb0);
      } else if constexpr (PREC == PREC_BF16X3) {
        lfrags_bf16(buf);
        compute_bf16();
      } else {
        mma_narrow(buf);
      }
    };
    static_assert(KCS >= 4 && KCS % 4 == 0, "staged GEMMs take K in {256, 512, 1024}");
    gload(st0, 0);
    gload(st1, 1);
    gload(st2, 2);
    gload(st3, 3);
    lstore(0, st0);
    if constexpr (4 < KCS) gload(st0, 4);
    __syncthreads();
    if constexpr (TRACE) ts[1] = clock_pinned();            // first chunk in LDS
#pragma unroll
    for (int kc = 0; kc < KCS; kc += 4) {
      if constexpr (TRACE) { if (kc == 4) ts[2] = clock_pinned(); }   // 4 chunks done
      mma(0);                                                       // chunk kc   (buffer 0)
      lstore(1, st1);
      if (kc + 5 < KCS) gload(st1, kc + 5);                         // kc is a constant after unrolling
      __syncthreads();
      mma(1);                                                       // chunk kc+1 (buffer 1)
      lstore(0, st2);
      if (kc + 6 < KCS) gload(st2, kc + 6);
      __syncthreads();
      mma(0);                                                       // chunk kc+2
      lstore(1, st3);
      if (kc + 7 < KCS) gload(st3, kc + 7);
      __syncthreads();
      mma(1);                                                       // chunk kc+3
      if (kc + 4 < KCS) {
        lstore(0, st0);
        if (kc + 8 < KCS) gload(st0, kc + 8);
      }
      __syncthreads();
    }
  }
  if (SPLIT) acc[0][0] += acc2;
  if constexpr (TRACE) ts[3] = clock_pinned();              // main loop done
  auto trace_out = [&]() {
    if constexpr (TRACE) {
      ts[4] = clock_pinned();                                // epilogue stores drained
      const long long wg = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
      if (lane == 0 && wg < 512) {
        unsigned long long* o = p.trace + (wg * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = ts[i];
        o[6] = rt0;
        o[7] = realtime_100mhz();
      }
    }
  };

  if constexpr (!LN) {
    // All epilogue loads are unconditional (clamped indices) and issued as one batch: a `cond ? load : 0`
    // becomes a branch + vmcnt(0) per load in hipcc's output (four serialised bias latencies before this).
    float bv[NREP];
    if (bias) {                        // uniform: one branch around the whole block of loads (raw split-K slabs have none)
#pragma unroll
      for (int b = 0; b < NREP; ++b) {
        const int col = n0 + b * 16 + r;
        bv[b] = bias[col < p.N ? col : p.N - 1];
      }
    } else {
#pragma unroll
      for (int b = 0; b < NREP; ++b) bv[b] = 0.f;
    }
    bool rowok[MREP][4];
    float rowmask[MREP][4];
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rowok[a][i] = m0 + a * 16 + g * 4 + i < p.M;
        rowmask[a][i] = 1.f;
      }
    if (p.lens) {                      // padded-frame zeroing (final_layer only): one block, loads batched
      int lim[MREP][4], tt[MREP][4];
#pragma unroll
      for (int a = 0; a < MREP; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = m0 + a * 16 + g * 4 + i;
          const int rc = row < p.M ? row : p.M - 1;
          const int grp = rc / p.rows_per_group;
          tt[a][i] = rc - grp * p.rows_per_group;
          lim[a][i] = p.lens[grp];
        }
#pragma unroll
      for (int a = 0; a < MREP; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowmask[a][i] = tt[a][i] < lim[a][i] ? 1.f : 0.f;
    }
    float* Cs = nullptr;
    if constexpr (STAGED) {
#if defined(MLDHIP_SIM)
      Cs = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
      extern __shared__ __attribute__((aligned(16))) float smem_epi[];
      Cs = smem_epi;
#endif
    }
    auto finish = [&](auto actfn) {
#pragma unroll
      for (int b = 0; b < NREP; ++b) {
        const int col = n0 + b * 16 + r;
        const bool colok = col < p.N;
#pragma unroll
        for (int a = 0; a < MREP; ++a)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = actfn(acc[a][b][i] + bv[b]) * rowmask[a][i];
            if constexpr (STAGED) {
              Cs[(wm * MREP * 16 + a * 16 + g * 4 + i) * (BN + 4) + wn * NREP * 16 + b * 16 + r] = v;
            } else {
              if (rowok[a][i] && colok) Y[(long long)(m0 + a * 16 + g * 4 + i) * p.ldy + col] = v;
            }
          }
      }
    };
    if (p.act == ACT_GELU) finish([](float x) { return gelu_erf(x); });
    else if (p.act == ACT_SILU) finish([](float x) { return silu(x); });
    else finish([](float x) { return x; });
    if constexpr (STAGED) {
      __syncthreads();
      store_tile_from_lds<BM, BN, WM * WN * 64>(Cs, Y, p.ldy, blockIdx.x * BM, blockIdx.y * BN, p.M, p.N, tid);
    }
    trace_out();
    return;
  } else {
  // ---------------- residual + LayerNorm epilogue (full rows live in this workgroup) -------------
  __shared__ float red[4][BM][WN];
  const int nstage = p.cvec ? 2 : 1;
  // ---- every epilogue operand is fetched up front, unconditionally (bias / residual / affine / cross-attn vector)
  float bv[NREP], gm1[NREP], bt1[NREP], gm2[NREP], bt2[NREP];
#pragma unroll
  for (int b = 0; b < NREP; ++b) {
    const int col = n0 + b * 16 + r;
    bv[b] = bias[col];
    gm1[b] = p.g1[col];
    bt1[b] = p.b1[col];
    gm2[b] = 1.f;
    bt2[b] = 0.f;
  }
  int rowc[MREP][4];
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + a * 16 + g * 4 + i;
      rowc[a][i] = row < p.M ? row : p.M - 1;
    }
  float resv[MREP][NREP][4];
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int b = 0; b < NREP; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) resv[a][b][i] = p.res[(long long)rowc[a][i] * p.ldres + n0 + b * 16 + r];
  float cvv[MREP][NREP][4];
  if (nstage == 2) {
#pragma unroll
    for (int b = 0; b < NREP; ++b) {
      gm2[b] = p.g2[n0 + b * 16 + r];
      bt2[b] = p.b2[n0 + b * 16 + r];
    }
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long grp = rowc[a][i] / p.rows_per_group;      // one division per row, not per element
#pragma unroll
        for (int b = 0; b < NREP; ++b) cvv[a][b][i] = p.cvec[grp * p.ldcvec + n0 + b * 16 + r];
      }
  }
  float vals[MREP][NREP][4];
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int b = 0; b < NREP; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) vals[a][b][i] = acc[a][b][i] + bv[b] + resv[a][b][i];
  const float inv_n = 1.0f / float(BN);
  for (int st = 0; st < nstage; ++st) {
    float mean[MREP][4], rstd[MREP][4];
    // pass 1: mean
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NREP; ++b) s += vals[a][b][i];
        s = sum16(s);
        if (r == 0) red[st * 2][wm * MREP * 16 + a * 16 + g * 4 + i][wn] = s;
      }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) s += red[st * 2][wm * MREP * 16 + a * 16 + g * 4 + i][w];
        mean[a][i] = s * inv_n;
      }
    // pass 2: variance about the mean (two-pass, like ATen's LayerNorm)
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < NREP; ++b) {
          float d = vals[a][b][i] - mean[a][i];
          s += d * d;
        }
        s = sum16(s);
        if (r == 0) red[st * 2 + 1][wm * MREP * 16 + a * 16 + g * 4 + i][wn] = s;
      }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) s += red[st * 2 + 1][wm * MREP * 16 + a * 16 + g * 4 + i][w];
        rstd[a][i] = rsqrtf(s * inv_n + kLnEps);
      }
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int b = 0; b < NREP; ++b) {
        const float gm = st == 0 ? gm1[b] : gm2[b], bt = st == 0 ? bt1[b] : bt2[b];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = (vals[a][b][i] - mean[a][i]) * rstd[a][i] * gm + bt;
          if (st == 0 && nstage == 2) v += cvv[a][b][i];
          vals[a][b][i] = v;
        }
      }
  }
  if constexpr (STAGED) {
#if defined(MLDHIP_SIM)
    float* Cs = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
    extern __shared__ __attribute__((aligned(16))) float smem_epi[];
    float* Cs = smem_epi;
#endif
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int b = 0; b < NREP; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          Cs[(wm * MREP * 16 + a * 16 + g * 4 + i) * (BN + 4) + wn * NREP * 16 + b * 16 + r] = vals[a][b][i];
    __syncthreads();
    store_tile_from_lds<BM, BN, WM * WN * 64>(Cs, Y, p.ldy, blockIdx.x * BM, 0, p.M, p.N, tid);
  } else {
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int b = 0; b < NREP; ++b) {
      const int col = n0 + b * 16 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + a * 16 + g * 4 + i;
        if (row < p.M) Y[(long long)row * p.ldy + col] = vals[a][b][i];
      }
    }
  }
  trace_out();
  }
}

}  // namespace mld
