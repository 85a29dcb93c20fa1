// Y = act(X · Wᵀ + b) on split-f16 MFMAs for the MFMA-bound shapes of the diffusion-only variant (config 4: M = 25 088 rows,
// K in {512, 1024}, N in {512, 1024, 1536}; mld_denoiser.py:208-221 -> cross_attention.py:195-233, 323-345): the big-tile,
// software-pipelined member of the staged GEMM family (gemm.hpp).
//
// Why a second staged kernel.  gemm_kernel's chunk loop is  barrier -> 16 ds_read_b128 -> 48 MFMA -> split + ds_write -> barrier:
// every wave of the workgroup is in the same phase at the same time, so the LDS phase (fragment reads of 8 waves: ~1 000 clocks),
// the matrix phase (1 536 clocks per chunk and SIMD at a 64 x 64 wave tile) and the store phase ADD UP -- r04 harness
// (tools/loopbench/gemm_bench.hip): 200-255 TFLOP/s at any tile shape from 64 x 128 to 256 x 128, i.e. <= 0.31 of the split-f16 roof.
// Here the fragments of chunk c + 1 are read WHILE chunk c is multiplied, into the registers chunk c has just released:
//   iteration c:  barrier;  for each row tile a: 3 NREP MFMAs (chunk c, fragments in registers) -> reload A[a] from buffer (c + 1) & 1;
//                 the last row tile goes column by column and reloads B[b] behind each column's three MFMAs;
//                 between the row tiles: split + ds_write of chunk c + 2 into buffer c & 1 (free: its fragments were read during
//                 iteration c - 1 and __syncthreads waits for them), global loads of chunk c + 2 + RD.
// ONE barrier per 32-wide K chunk, LDS reads / writes and the global prefetch ring all run under the matrix instructions, no second
// fragment set (the reload reuses the registers), RD chunks of global loads in flight across barriers (workgroup-scope fences do not
// drain vmcnt outside tgsplit mode).
//
// Operands: X fp32 [M][K] (split into half planes while it is written to LDS, like gemm_kernel's PREC_BF16X3 path); W the pre-split
// image of the weight arena (elementwise.hpp split_bf16_weights_kernel; GemmArgs::w_split must be set).  Output through an LDS tile
// with 16-byte stores (gemm.hpp store_tile_from_lds).
//
// Workgroup -> tile map (XCD aware): dispatch order id lands on XCD id % 8 (MI355X_MICROARCH.md); XCD x walks row tiles x, x + 8, ...
// and, inside a row tile, all column tiles back to back, so the 2-6 workgroups that read the same X panel run on ONE XCD at about
// the same time (its L2 serves the re-reads; the weight image, 1-3 MB, stays resident in every L2).
#pragma once
#include "gemm.hpp"

namespace mld {

// Placement pins.  hipcc linearises a basic block with a register-pressure list scheduler BEFORE the machine scheduler sees the
// sched_barrier fences: nodes without a chain (matrix instructions, the VALU split of a staged row) float to wherever their operands
// become available -- the split of chunk c + 5 directly behind its global load (s_waitcnt vmcnt(0): the prefetch ring collapses), all
// matrix instructions of an iteration in front of its fragment reloads (ten ds_read_b128 bunched in front of the barrier).  An empty
// asm volatile that "rewrites" a register is chained like the fences, so whatever consumes the register stays behind it.
__device__ __forceinline__ void pin4(U4& v) {
#if !defined(MLDHIP_SIM)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t t = __builtin_bit_cast(u32x4_t, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(U4, t);
#endif
}
__device__ __forceinline__ void pin4(F4& v) {
#if !defined(MLDHIP_SIM)
  f32x4 t = __builtin_bit_cast(f32x4, v);
  asm volatile("" : "+v"(t));
  v = __builtin_bit_cast(F4, t);
#endif
}

#ifndef GP_EXP
#define GP_EXP 0      // tools/loopbench experiments (measurement builds with WRONG results; 0 in the library): 1 = fragments are not reloaded,
                      // 2 = no LDS stores in the loop, 4 = no global loads in the loop, 8 = no matrix instructions, 16 = no barrier in the loop
#endif

template <int WM, int WN, int MREP, int NREP>
constexpr int gemm_pipe_lds_bytes() { return gemm_lds_bytes<WM, WN, MREP, NREP>(); }

// 1-D grid size: 8 * ceil(row tiles / 8) * column tiles (the surplus workgroups of the last round exit at once)
template <int WM, int WN, int MREP, int NREP>
inline unsigned gemm_pipe_grid(int M, int N) {
  constexpr int BM = WM * MREP * 16, BN = WN * NREP * 16;
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  return (unsigned)(((mt + 7) / 8) * 8 * nt);
}

template <int WM, int WN, int MREP, int NREP, int KCS, int RD = 3>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_pipe_x3_kernel(GemmArgs p) {
  constexpr int BM = WM * MREP * 16, BN = WN * NREP * 16, NT = WM * WN * 64, ROWS = BM + BN;
  constexpr int NLD = ROWS * 8 / NT, NLA = BM * 8 / NT;
  static_assert(ROWS * 8 % NT == 0 && BM * 8 % NT == 0, "panel rows must tile the workgroup");
  static_assert(KCS >= 4 && RD >= 2 && RD <= 4, "chunk pipeline: K >= 128, 2-4 chunks of global loads in flight");
  static_assert(MREP >= 3, "the pipeline spreads its LDS stores / global loads over the slots of row tiles 0 .. MREP - 2");
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tm = (kk / nt) * 8 + xcd, tn = kk % nt;
  if (tm >= mt) return;
  const int bm0 = tm * BM, bn0 = tn * BN;
#if defined(GP_TRACE)   // tools/loopbench only: shader-clock stamps of wave 0 (start, first barrier, loop end, tile in LDS, stores issued, stores drained) + 100 MHz wall clock
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long rt0 = realtime_100mhz();
  ts[0] = clock_light();
#define GP_STAMP(k) ts[k] = clock_light()
#else
#define GP_STAMP(k)
#endif
  const float* A = p.A;
  const float* W = p.W;

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int a = 0; a < MREP; ++a)
#pragma unroll
    for (int b = 0; b < NREP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's bias values, requested first (a load in front of the epilogue costs it an L2 round trip: r04 phase stamps); N % BN == 0
  F4 bv[NREP];
  if (p.bias) {
#pragma unroll
    for (int b = 0; b < NREP; ++b) bv[b] = ld4(p.bias + bn0 + wn * NREP * 16 + b * 16 + g * 4);
  } else {
#pragma unroll
    for (int b = 0; b < NREP; ++b) bv[b] = F4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- global -> register staging: slot j of a thread is the 16-byte piece (idx & 7) of panel row idx >> 3, idx = tid + j NT
  F4 st[RD][NLD];
  const float* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
    if (j < NLA) {
      int m = bm0 + row;
      m = m < p.M ? m : p.M - 1;
      src[j] = A + (long long)m * p.lda + c4 * 4;
    } else {
      int n = bn0 + row - BM;
      n = n < p.N ? n : p.N - 1;
      src[j] = W + (long long)n * p.ldw + c4 * 4;
    }
  }
  auto gload = [&](F4 (&s)[NLD], int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) s[j] = ld4(src[j] + kc * 32);
  };
  // A rows: split into (high | low) half planes; W rows: already the row image
  auto lstore_a = [&](int buf, F4 (&s)[NLD]) __attribute__((always_inline)) {
    float* dst = smem + buf * ROWS * kGemmLdsStride;
#pragma unroll
    for (int j = 0; j < NLA; ++j) pin4(s[j]);           // the split happens HERE, not behind the load
#pragma unroll
    for (int j = 0; j < NLA; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
      unsigned h0, l0, h1, l1;
      split16_pair(s[j].x, s[j].y, h0, l0);
      split16_pair(s[j].z, s[j].w, h1, l1);
      unsigned* rowp = reinterpret_cast<unsigned*>(dst + row * kGemmLdsStride);
      *reinterpret_cast<uint2_t*>(rowp + c4 * 2) = uint2_t{h0, h1};
      *reinterpret_cast<uint2_t*>(rowp + 16 + c4 * 2) = uint2_t{l0, l1};
    }
  };
  auto lstore_w = [&](int buf, const F4 (&s)[NLD]) __attribute__((always_inline)) {
    float* dst = smem + buf * ROWS * kGemmLdsStride;
#pragma unroll
    for (int j = NLA; j < NLD; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
      st4(dst + row * kGemmLdsStride + c4 * 4, s[j]);
    }
  };
  // ---- fragments: lane (r, g) holds k = 8g .. 8g + 7 of row r of a tile as packed halves: words 4g .. 4g + 3 of the high plane,
  // 16 + 4g .. of the low plane (row stride 40 words = 8 mod 16: conflict-free ds_read_b128, gemm.hpp kGemmLdsStride)
  U4 ahi[MREP], alo[MREP], bhi[NREP], blo[NREP];
  const int aoff = (wm * MREP * 16 + r) * kGemmLdsStride + g * 4, boff = (BM + wn * NREP * 16 + r) * kGemmLdsStride + g * 4;
  auto load_a = [&](int buf, int t) __attribute__((always_inline)) {
    const float* q = smem + buf * ROWS * kGemmLdsStride + aoff + t * 16 * kGemmLdsStride;
    ahi[t] = *reinterpret_cast<const U4*>(q);
    alo[t] = *reinterpret_cast<const U4*>(q + 16);
  };
  auto load_b = [&](int buf, int t) __attribute__((always_inline)) {
    const float* q = smem + buf * ROWS * kGemmLdsStride + boff + t * 16 * kGemmLdsStride;
    bhi[t] = *reinterpret_cast<const U4*>(q);
    blo[t] = *reinterpret_cast<const U4*>(q + 16);
  };

  auto mma = [&](const U4& x, const U4& y, f32x4 c) __attribute__((always_inline)) {
    if constexpr ((GP_EXP & 8) != 0) { c[0] += __builtin_bit_cast(float, x.x ^ y.w); return c; }
    else return mfma_x3_16x16x32(x, y, c);
  };
  // ---- prologue: chunks 0 and 1 in LDS, chunks 2 .. RD + 1 in flight, fragments of chunk 0 in registers
#pragma unroll
  for (int j = 0; j < RD; ++j)
    if (j < KCS) gload(st[j], j);
  lstore_a(0, st[0]);
  lstore_w(0, st[0]);
  if (RD < KCS) gload(st[0], RD);
  lstore_a(1, st[1 % RD]);
  lstore_w(1, st[1 % RD]);
  if (RD + 1 < KCS) gload(st[1 % RD], RD + 1);
  __syncthreads();
  GP_STAMP(1);
#pragma unroll
  for (int t = 0; t < MREP; ++t) load_a(0, t);
#pragma unroll
  for (int t = 0; t < NREP; ++t) load_b(0, t);

  // One iteration = MREP row tiles.  Row tiles 0 .. MREP - 2 are three groups of NREP matrix instructions each; behind every group sits a
  // SLOT with a few instructions of the other pipes (hand interleave: hipcc clusters them otherwise and a cluster of 8-12 VALU
  // instructions holds the issue port for longer than the 12 free clocks behind a matrix instruction):
  //   slots 0 ..: one pair of a staged X piece each -- 5 VALU (split16_two: values produced by the kernels upstream are not
  //   range-clamped, rt.hpp) and, behind a piece's second pair, its two 8-byte LDS stores; then the W pieces' 16-byte stores; then the
  //   global loads of chunk c + 2 + RD; the slot behind a row tile's last group reloads that tile's fragments.
  constexpr int NS = 3 * (MREP - 1), NPAIR = 2 * NLA, NWS = NLD - NLA;
  constexpr int S_W0 = NPAIR < NS - 4 ? NPAIR : NS - 4;                         // first slot with W stores
  constexpr int WPS = (NWS + 1) / 2, GPS = (NLD + 1) / 2;                        // W stores over two slots, global loads over the last two
  constexpr int PPS = (NPAIR + S_W0 - 1) / S_W0;                                 // pairs per slot
  unsigned hh[NLA][2], ll[NLA][2];
#pragma unroll
  for (int c = 0; c < KCS; ++c) {
    const int nb = (c + 1) & 1, fb = c & 1;          // buffer of chunk c + 1 (read) / of chunk c, free from here on (written: chunk c + 2)
    const bool more = c + 1 < KCS && !(GP_EXP & 1), stage = c + 2 < KCS && !(GP_EXP & 2), fetch = c + 2 + RD < KCS && !(GP_EXP & 4);
    F4 (&sg)[NLD] = st[(c + 2) % RD];
    // chunk c + 1 is complete in LDS and everybody's fragment reads of chunk c are done (__syncthreads waits for the issuing wave's LDS reads
    // first).  Also in front of iteration 0: its first slot overwrites buffer 0 while a slower wave could still be reading its fragments of
    // chunk 0 behind the prologue's barrier -- a race that showed as run-to-run differences of 1e-2 on the joints of a 20-step sample (r04).
    if (!(GP_EXP & 16)) __syncthreads();
    auto slot = [&](int k) __attribute__((always_inline)) {
      if (stage) {
        float* dst = smem + fb * ROWS * kGemmLdsStride;
#pragma unroll
        for (int q = k * PPS; q < (k + 1) * PPS && q < NPAIR; ++q) {
          if (k >= S_W0) break;
          const int j = q >> 1;
          if ((q & 1) == 0) {
            pin4(sg[j]);                                                  // the split happens HERE, not behind the load
            split16_two(sg[j].x, sg[j].y, hh[j][0], ll[j][0]);
          } else {
            split16_two(sg[j].z, sg[j].w, hh[j][1], ll[j][1]);
            const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
            unsigned* rowp = reinterpret_cast<unsigned*>(dst + row * kGemmLdsStride);
            *reinterpret_cast<uint2_t*>(rowp + c4 * 2) = uint2_t{hh[j][0], hh[j][1]};
            *reinterpret_cast<uint2_t*>(rowp + 16 + c4 * 2) = uint2_t{ll[j][0], ll[j][1]};
          }
        }
        if (k >= S_W0 && k < S_W0 + 2) {
#pragma unroll
          for (int j = NLA + (k - S_W0) * WPS; j < NLA + (k - S_W0 + 1) * WPS && j < NLD; ++j) {
            const int idx = tid + j * NT, row = idx >> 3, c4 = idx & 7;
            st4(dst + row * kGemmLdsStride + c4 * 4, sg[j]);
          }
        }
      }
      if (fetch && k >= NS - 2) {
#pragma unroll
        for (int j = (k - (NS - 2)) * GPS; j < (k - (NS - 2) + 1) * GPS && j < NLD; ++j) sg[j] = ld4(src[j] + (c + 2 + RD) * 32);
      }
    };
#pragma unroll
    for (int a = 0; a + 1 < MREP; ++a) {
      pin4(ahi[a]);
      pin4(alo[a]);
#pragma unroll
      for (int b = 0; b < NREP; ++b) acc[a][b] = mma(bhi[b], alo[a], acc[a][b]);
      slot(3 * a);
      sched_fence();
#pragma unroll
      for (int b = 0; b < NREP; ++b) acc[a][b] = mma(blo[b], ahi[a], acc[a][b]);
      slot(3 * a + 1);
      sched_fence();
#pragma unroll
      for (int b = 0; b < NREP; ++b) acc[a][b] = mma(bhi[b], ahi[a], acc[a][b]);
      if (more) load_a(nb, a);
      slot(3 * a + 2);
      sched_fence();
    }
    {
      constexpr int a = MREP - 1;
#pragma unroll
      for (int b = 0; b < NREP; ++b) {
        pin4(bhi[b]);
        pin4(blo[b]);
        acc[a][b] = mma(bhi[b], alo[a], acc[a][b]);
        acc[a][b] = mma(blo[b], ahi[a], acc[a][b]);
        acc[a][b] = mma(bhi[b], ahi[a], acc[a][b]);
        if (more) load_b(nb, b);
        sched_fence();
      }
      if (more) load_a(nb, a);
      sched_fence();
    }
  }

  GP_STAMP(2);
  // ---- epilogue: bias + activation, through an LDS tile, 16-byte stores.  The products are taken TRANSPOSED (weights as the MFMA's A
  // operand): lane (r, g) holds row r of a tile and four CONSECUTIVE columns 4g .. 4g + 3, so a tile enters the LDS image as one
  // 16-byte store per lane (the plain layout needs four 4-byte ones: r04 phase stamps, 3 400 of a workgroup's 45 000 clocks at K = 512)
  __syncthreads();                                    // everybody is done with the chunk buffers
  float* Cs = smem;
  auto finish = [&](auto actfn) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < MREP; ++a)
#pragma unroll
      for (int b = 0; b < NREP; ++b)
        st4(Cs + (wm * MREP * 16 + a * 16 + r) * (BN + 4) + wn * NREP * 16 + b * 16 + g * 4,
            F4{actfn(acc[a][b][0] + bv[b].x), actfn(acc[a][b][1] + bv[b].y), actfn(acc[a][b][2] + bv[b].z), actfn(acc[a][b][3] + bv[b].w)});
  };
  if (p.act == ACT_GELU) finish([](float x) { return gelu_erf(x); });
  else if (p.act == ACT_SILU) finish([](float x) { return silu(x); });
  else finish([](float x) { return x; });
  __syncthreads();
  GP_STAMP(3);
  store_tile_from_lds<BM, BN, NT>(Cs, p.Y, p.ldy, bm0, bn0, p.M, p.N, tid);
#if defined(GP_TRACE)
  GP_STAMP(4);
  ts[5] = clock_pinned();
  if (tid == 0 && p.trace) {
    unsigned long long* o = p.trace + (long long)blockIdx.x * 8;
    for (int i = 0; i < 6; ++i) o[i] = ts[i];
    o[6] = rt0;
    o[7] = realtime_100mhz();
  }
#endif
#undef GP_STAMP
}

}  // namespace mld
