// The whole reverse-diffusion loop of the latent models as ONE persistent launch ("sample-major" loop).
//
// Why it exists.  In the denoiser (mld_denoiser.py:135-228) a motion only ever meets its own three tokens [latent, time, text]
// (the 3-token self-attention of cross_attention.py:259-272), its own unconditional / conditional pair (classifier-free guidance,
// mld.py:339-342) and its own latent (the DDIM step, mld.py:345-346): motions never interact.  The launch-per-GEMM families
// (tile32.hpp, strip.hpp) nevertheless send every activation through HBM / L2 41 times per step and pay 2 050 dependent launches
// per call, because they split each GEMM over the CUs by output COLUMNS.  Here the work is split by MOTIONS instead: a workgroup
// owns 8 motions = 16 rows of the CFG batch x 3 tokens = 48 token rows for the whole call, keeps them in LDS, and streams the
// weights: 1 launch, no inter-workgroup traffic, no activation ever leaves the CU (except the four skip activations, parked in a
// workgroup-private slice of HBM):
//   per reverse step and workgroup  48 rows x 7.6 M weights x 2 = 730 MFLOP  against  30.4 MB of weights streamed from L2 / MALL
//   = 24 FLOP per byte.  On exact-fp32 MFMAs (256 FLOP/clk/CU) the stream needs 10.7 B/clk/CU, well under what the register-ring
//   + ds_write staging sustains: MFMA bound (measured r03: 77 ms for 50 steps at ANY batch up to 2 048 motions = 0.77 of the fp32
//   MFMA peak).  On split-f16 MFMAs (rt.hpp: 3 instead of 8 matrix instructions per chunk, each twice as fast) the matrix work
//   shrinks 5x: 27 ms, of which the matrix pipe is busy ~45 %.  Phase counters of the kernel itself (fused_dbg 5, tools/trace_loop.py,
//   profiles/r03_loop_phase_trace.json): the product-only phases (QKV, out-projection, skip linears) run at 82-84 % of the matrix
//   rate; the feed-forward phase, whose GELU / split / hidden-image stores and eight barriers share the issue ports with its matrix
//   instructions, at 56 %; attention scores + softmax, the two LayerNorm phases and the skip handling -- no matrix work at all -- take
//   a quarter of the time.  The epilogues of a phase cannot overlap the next phase's products, which depend on them.
// It wins once a call carries enough motions to give most CUs a workgroup (2 048 motions = 256 workgroups = one per CU); below
// a few hundred motions the column-split families finish sooner (path_latent.hpp use_fused).
//
// Row order inside the workgroup: row = 16 t + c, t = token, c = row of the CFG batch (c < 8: unconditional half of the 8
// motions, c >= 8: conditional half).  A 16-row MFMA tile is then ONE token of all 16 CFG rows.  Every product is taken TRANSPOSED
// (weights as the A operand, the rows as B): the accumulator registers of lane (r, g) of wave w hold, for each of the three row
// tiles, row c = r and columns 16 w + 4 g .. + 3 of a 128-column block -- the three tokens of the SAME CFG row and the same four
// consecutive columns.  The 3-token attention, the softmax and P.V are in-lane arithmetic on accumulators (the 64-column dot products
// are four in-lane terms, two shuffles over g and one LDS exchange among the four waves of a head; a softmax is worked out by 4 lanes
// per row), q, k, v never exist in memory, a LayerNorm's row statistics are eight in-lane terms and two shuffles per tile, and a
// row's elements enter an operand image as one 8-byte store per plane.  (The first form of this kernel kept the plain accumulator
// layout -- rows 4g + i, one column per lane: sixteen-lane DPP reductions per element, softmaxes repeated in 16 lanes, 2-byte
// image stores; attention 4.2 -> 2.6 ms, loop 29.7 -> 26.7 ms with the transposed one.)  Residuals are re-read from the operand
// image they were multiplied from (high + low half, the value the GEMMs saw), so no activation stays in registers across a phase.
//
// Weight stream: a weight element is used by exactly ONE wave (wave w owns columns 16w .. 16w + 15 of every 128-column block), so
// staging weights through LDS would buy nothing and cost a store, a read and a barrier per item (the first build of this kernel
// did: 42 ms per call in split mode, bound by LDS traffic and 1 856 barriers per step).  Instead `finalize` re-packs the GEMM
// weights into "items" of 128 output columns x 32 k in FRAGMENT order -- [wave][lane][8 words]: lane (r, g) of wave w finds the
// two 16-byte MFMA operands of weight row 16w + r (fp32: k-slots 4g .. 4g + 3 and 16 + 4g .. + 3; split: high and low halves of
// k = 8g .. 8g + 7) contiguous -- in the order the kernel consumes them, and every lane loads its 32 bytes per item straight
// into an 8-deep register ring: 2 KB contiguous per wave and item, no LDS, no barrier.  Waves only meet where activations
// change hands (17 barriers per layer).  Items that multiply the same 32 columns of A are adjacent in the stream ([Q, K, V] of a
// chunk; both column blocks of the out-projection, linear2 and the skip linears; two hidden blocks of linear1), so an A fragment
// read from LDS feeds 2-3 items.  The sequence is uniform over phases, layers and steps (the first items of a step are repeated
// behind its last one, so the look-ahead needs no wrap test): the prefetch never drains.  Bound in split mode: the L2 -> CU path (16 KB per item and CU at 64 B/clk) next to 9 MFMAs per item and wave.
//
// Replaces, per step: mld_denoiser.py:143-228 (token assembly, SkipTransformerEncoder, final norm), mld.py:325-346 (CFG + DDIM).
#pragma once
#include "tile32.hpp"

namespace mld {

// packed biases / LayerNorm parameters (floats): per layer [in_b 768 | out_b 256 | n1_w 256 | n1_b 256 | l1_b 1024 | l2_b 256 | n2_w 256 | n2_b 256],
// then skip-linear biases [nb][256], encoder.norm weight / bias, query_pos.pe[0]
constexpr int kLsInB = 0, kLsOutB = 768, kLsN1W = 1024, kLsN1B = 1280, kLsL1B = 1536, kLsL2B = 2560, kLsN2W = 2816, kLsN2B = 3072, kLsLayer = 3328;

// items of one layer: QKV 48, out-proj 16, feed-forward 128; a skip linear: 32
constexpr int kLoopItemsLayer = 192, kLoopItemsSkip = 32, kLoopItemFloats = 128 * 32;

struct LoopItem { long long src; int ld; int pad; };   // element [row0][k0] of a weight (floats into the arena), row stride

struct LoopArgs {
  const float* stream;     // [ips + 8][8 waves][64 lanes][8 words] weight items in consumption order, fragment layout (fp32 or split halves); the last 8 repeat the first 8
  int ips;                 // items per reverse step
  const float* small;      // packed small parameters (layout above)
  const float* T1;         // [n][256] time-token rows (time MLP + pe[1]) of the scheduler's timesteps
  const float* TP;         // [2B][256] condition-token rows (+ pe[2]), unconditional half first
  const float* init_lat;   // [B][256]
  float* lat;              // [B][256] latents after the last step
  float* skip;             // [gridDim.x][nb][48][256] workgroup-private skip activations
  const float* ddim;       // [n][4] DdimCoef per step
  int B, L, n;
  float guidance, init_sigma;
  unsigned long long* trace = nullptr;   // DBG 5 (measurement build): [workgroup < 64][wave][8] shader cycles per phase, summed over steps and layers
};

#ifndef LF_EXP
#define LF_EXP 0          // tools/loopbench experiments (measurement builds with WRONG results; 0 in the library): 1 / 4 = linear1 / linear2 re-use stale A
                          // fragments (no LDS reads), 2 = no barrier per hidden block, 8 = the weight ring is never refreshed, 16 = matrix instructions and
                          // fragment reads of TWO row tiles per item instead of three (the matrix work of a 5-motion workgroup: VERDICT r4 item 3's upper bound)
#endif
constexpr int kLfMmaTiles = (LF_EXP & 16) ? 2 : 3;
#ifndef LF_RING
#define LF_RING 8         // weight items in flight per lane (the stream carries 8 look-ahead items behind a step: 4 or 8; tools/loopbench A/B)
#endif
#ifndef LF_PIN
#define LF_PIN 2          // tools/loopbench A/B: 0 = no placement pins, 1 = the accumulators of EVERY item are pinned behind its matrix instructions
                          // (pin_acc below), 2 (the library) = only in the skip linears, where the matrix instructions sank
#endif
constexpr int kLfXs = 264;                                // LDS row stride (words), = 8 mod 16: conflict-free fragment reads (strip.hpp)
constexpr int kLfHs = 136;                                // ... of a 128-wide block of the hidden activation
constexpr int kLfXFloats = 48 * kLfXs, kLfHFloats = 48 * kLfHs, kLfAFloats = 2 * kLfHFloats;     // As: the attention output [48][264], or two hidden blocks
constexpr int kLfScFloats = 2 * 8 * 144, kLfRedFloats = 2 * 8 * 48, kLfLatFloats = 8 * 256;
constexpr int kLfPrmFloats = kLsLayer + 256;        // a layer's packed small parameters + the bias of the skip linear behind it (if any)
static_assert(kLfAFloats >= kLfXFloats, "the attention output and the two hidden-block buffers share one region");
constexpr int kLoopLdsBytes = (kLfXFloats + kLfAFloats + kLfScFloats + kLfRedFloats + kLfLatFloats + 2 * kLfPrmFloats) * 4;   // 152 064 B: one workgroup per CU


// finalize-time: gathers the weight items into consumption order and fragment layout.  Thread (w, r, g) of item i writes the 8
// words lane (r, g) of wave w will load: fp32: W[16w + r][4g .. 4g + 3], W[16w + r][16 + 4g .. + 3] of the item's 32 k; X3: the packed
// high halves of k = 8g .. 8g + 7 (4 words), then the low halves.  grid = items, block = 512.
template <bool X3>
__global__ __launch_bounds__(512) void pack_loop_stream_kernel(const float* __restrict__ arena, const LoopItem* __restrict__ items,
                                                               float* __restrict__ out) {
  const LoopItem it = items[blockIdx.x];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const float* src = arena + it.src + (long long)(16 * w + r) * it.ld;
  float* dst = out + (long long)blockIdx.x * kLoopItemFloats + threadIdx.x * 8;
  if constexpr (!X3) {
    st4(dst, ld4(src + 4 * g));
    st4(dst + 4, ld4(src + 16 + 4 * g));
  } else {
    const F4 a = ld4(src + 8 * g), b = ld4(src + 8 * g + 4);
    U4 hi, lo;
    split16_pair(a.x, a.y, hi.x, lo.x);
    split16_pair(a.z, a.w, hi.y, lo.y);
    split16_pair(b.x, b.y, hi.z, lo.z);
    split16_pair(b.z, b.w, hi.w, lo.w);
    *reinterpret_cast<U4*>(dst) = hi;
    *reinterpret_cast<U4*>(dst + 4) = lo;
  }
}

// grid = ceil(B / 8), block = 512 (8 waves, two per SIMD).  X3 = false: exact-fp32 MFMAs; true: split-f16 MFMAs.
// kLoopRing = items in flight per lane (8 VGPRs each; every group of items is a multiple of 8 long).
// DBG: 0 = the product build; 5 = the same arithmetic with phase counters (mldhip_set_option "fused_dbg" 5, tools/trace_loop.py).  1-4 are
// measurement builds with WRONG results that only tools/loopbench instantiates -- they are not in libmldhip.so: 1 = the weight ring is
// loaded once and never refreshed, 2 = the stream is loaded but not multiplied, 3 = identity instead of GELU, 4 = no feed-forward epilogue.
// SWZ (split mode, always on): the 16-byte groups of an operand row are stored XOR-swizzled by the row -- physical word = logical word ^ 4 ((row >> 2) & 3),
// i.e. group (g ^ (r >> 2)) of each 16-word half chunk.  The image's 8-byte row stores (put_row: sixteen rows 264 = 8 mod 32 words
// apart per 16-lane group) hit each bank pair four times (ds_write_b64 is served per 16 contiguous lanes over 32 banks); swizzled,
// twice -- the best an 8-byte store at this stride can do -- while the fragments' ds_read_b128 stay conflict free (slot
// 2 r + (g ^ (r >> 2)) mod 16 is still a permutation inside each of the instruction's four lane groups; tests/test_lds_layout.py).
template <bool X3, int DBG = 0>
__global__ __launch_bounds__(512, 2) void den_loop_kernel(LoopArgs p) {
  constexpr int kLoopRing = LF_RING;      // items in flight per lane.  (r03: 8 spilled ring slots around the epilogues at the 256-register cap and lost, 33.1 vs
                                          // 29.3 ms; with the skip linears pinned -- pin_acc: 194 registers, no scratch -- 8 wins: 19.07 / 18.65 ms at 1 280 motions for
                                          // no pins + ring 4 / pins + ring 8, 25.25 / 24.99 ms at 2 048, profiles/r04_loop_experiments.json)
  constexpr bool SWZ = X3;          // split images are always row-swizzled (r03c: 27.15 -> 26.75 ms; r04a: LDS bank-conflict cycles 1.43e9 -> 0.59e9 per launch)
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* Xs = smem;                       // [48][264] layer input / norm1 output (the A operand; fp32, or the split image)
  float* As = Xs + kLfXFloats;            // [48][264] attention output, then [2][48][136]: two 128-wide blocks of the hidden activation in turn
  float* sc = As + kLfAFloats;            // [2 head pairs][2 heads of the pair][9 (t, u)][16 rows][4 waves of the head] partial attention scores
  float* red = sc + kLfScFloats;          // [2 passes][48 rows][8 waves] LayerNorm partial sums
  float* lats = red + kLfRedFloats;       // [8][256] the workgroup's latents
  float* prm = lats + kLfLatFloats;       // [2][kLfPrmFloats] the current / the next layer's biases and LayerNorm parameters (see prm_fetch)
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int s0 = blockIdx.x * 8, nb = (p.L - 1) / 2;
  const float* sm_skip = p.small + (long long)p.L * kLsLayer;
  const float* sm_fin = sm_skip + nb * 256;

  // ---- weight ring: this lane's two MFMA operands (32 bytes) of the next kLoopRing items, straight from the fragment-ordered stream
  // DBG 5: cycles per phase (0 QKV products, 1 scores + softmax + attention output, 2 out-projection, 3 residual + norm1, 4 feed-forward,
  // 5 residual + norm2 + skip handling, 6 end of step), each stamp behind s_waitcnt 0 -- the counters perturb the overlap they measure
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tph = 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr (DBG == 5) {
      const unsigned long long t = clock_pinned();
      ph[k] += t - tph;
      tph = t;
    }
  };
  // The stream holds ips + kLoopRing items: the first kLoopRing of a step once more behind the last, so the ring's look-ahead
  // runs off the end of a step without a wrap test per load; the pointer goes back to item kLoopRing at the top of every step
  // (the ring then holds items 0 .. kLoopRing - 1, loaded through the copies).
  unsigned goff = (unsigned)tid * 8u;       // this lane's word offset into the stream (< 2^31: the stream is ~30 MB)
  F4 ring[kLoopRing][2];
  auto gload = [&](int slot) __attribute__((always_inline)) {
    const float* s = p.stream + goff;
    ring[slot][0] = ld4(s);            // default cache policy: the CUs of an XCD re-read the same items from its L2 (`nt` loads: 19.0 -> 35.8 ms, r04)
    ring[slot][1] = ld4(s + 4);
    goff += (unsigned)kLoopItemFloats;
#if !defined(MLDHIP_SIM)
    asm volatile("" : "+v"(goff));     // one running offset: without this hipcc rewrites every load as base + constant, computes the
                                       // addresses of a whole phase ahead of its matrix instructions and spills them (r03: 2.5x on the skip
                                       // linears).  (An opaque POINTER loses its address space: flat loads, which count on lgkmcnt.)
#endif
  };
  // Items are numbered j = 0 .. inside a group (every group is a multiple of kLoopRing items); item j sits in ring slot j % kLoopRing
  // and is replaced by item j + kLoopRing as soon as its MFMAs are issued.
  // One item: this wave's 16 weight rows (= output columns) x 32 k against the three row tiles, whose fragments the caller read
  // from LDS: both operand formats keep a chunk of a row as 32 words read as words 4g .. 4g + 3 and 16 + 4g .. + 3.
  // Every product is taken TRANSPOSED (weights as the A operand, the token rows as B -- the same fragment registers in the other
  // argument): lane (r, g) of wave w then holds row r of a tile, columns 16 w + 4 g .. + 3 of a 128-column block, i.e. four
  // CONSECUTIVE elements of one row.  Everything around the products is cheaper in that layout: a row's statistics (LayerNorm,
  // attention scores) are four in-lane terms and two lane shuffles over g instead of sixteen-lane reductions per element, a softmax
  // is worked out by 4 lanes per row instead of 16, and a row's elements reach an operand image as one 8-byte store per plane instead
  // of eight 2-byte ones (r03 phase counters, plain layout: attention 4.2 ms, the two LayerNorm phases 6.8 ms of the loop's 29.8).
  // hipcc linearises a basic block with a register-pressure list scheduler BEFORE the machine scheduler sees the fences: matrix
  // instructions have no chain and may sink below the (chained) loads of several later items -- in the skip linears they did: the A
  // fragments and ring slots of five chunks were live at once and spilled (rounds 3-4: 176-216 B of scratch per lane at 256 registers, every
  // reload draining the in-order memory counter).  An empty asm volatile that "rewrites" the accumulators is chained like the fences: the
  // item's matrix instructions stay in front of it.  Pinned skip linears: 194 registers, NO scratch.  (Pinning every item is level; pinning
  // the nine GELU slices of mma_item_sliced into their slots -- hipcc collects them behind one matrix instruction -- gives the intended
  // 1 MFMA : 4 VALU interleave in the machine code and is 2 % SLOWER: the feed-forward phase is not an issue-order effect.)
  auto pin_acc = [](f32x4 (&a)[3]) __attribute__((always_inline)) {
#if !defined(MLDHIP_SIM)
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
#endif
  };
  auto mma_item = [&](int j, const F4 (&x)[3][2], f32x4 (&acc)[3], bool pin = false) __attribute__((always_inline)) {
    const int slot = j % kLoopRing;
    if constexpr (DBG == 2) {
      acc[0][0] += ring[slot][0].x + ring[slot][1].w + x[0][0].x;
    } else if constexpr (X3) {
      const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
#pragma unroll
      for (int t = 0; t < kLfMmaTiles; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][1]), acc[t]);
#pragma unroll
      for (int t = 0; t < kLfMmaTiles; ++t) acc[t] = mfma_x3_16x16x32(wl, __builtin_bit_cast(U4, x[t][0]), acc[t]);
#pragma unroll
      for (int t = 0; t < kLfMmaTiles; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][0]), acc[t]);
    } else {
      const F4 y0 = ring[slot][0], y1 = ring[slot][1];
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y0.x, x[t][0].x, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y0.y, x[t][0].y, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y0.z, x[t][0].z, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y0.w, x[t][0].w, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y1.x, x[t][1].x, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y1.y, x[t][1].y, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y1.z, x[t][1].z, acc[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(y1.w, x[t][1].w, acc[t]);
    }
    if ((LF_PIN & 1) || ((LF_PIN & 2) && pin)) pin_acc(acc);
    if constexpr (DBG != 1 && !(LF_EXP & 8)) gload(slot);
    sched_fence();
  };
  // An item with an epilogue threaded through it: `epi(k)`, k = 0 .. 8, is a few VALU / LDS instructions of work that does
  // not depend on this item; slice k is issued right behind the item's k-th group of matrix instructions and a scheduling fence pins
  // the pair.  MFMA and the other VALU instructions share a SIMD's issue port but not its pipes: behind every 16-cycle matrix
  // instruction there are three free issue slots, and an in-order wave fills them only if the next instructions in ITS stream are
  // not matrix instructions.  (r03 phase counters: the feed-forward phase took 14.2 ms of the loop's 29.8 where its matrix
  // instructions need 7.9 -- the compiler's own interleave left MFMA and GELU time adding up.)
  auto mma_item_sliced = [&](int j, const F4 (&x)[3][2], f32x4 (&acc)[3], auto&& epi) __attribute__((always_inline)) {
    const int slot = j % kLoopRing;
    if constexpr (DBG == 2) {
      acc[0][0] += ring[slot][0].x + ring[slot][1].w + x[0][0].x;
#pragma unroll
      for (int k = 0; k < 9; ++k) epi(k);
    } else if constexpr (X3) {
      const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
#pragma unroll
      for (int t = 0; t < 3; ++t) { if (t < kLfMmaTiles) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][1]), acc[t]); epi(t); sched_fence(); }
#pragma unroll
      for (int t = 0; t < 3; ++t) { if (t < kLfMmaTiles) acc[t] = mfma_x3_16x16x32(wl, __builtin_bit_cast(U4, x[t][0]), acc[t]); epi(3 + t); sched_fence(); }
#pragma unroll
      for (int t = 0; t < 3; ++t) { if (t < kLfMmaTiles) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][0]), acc[t]); epi(6 + t); sched_fence(); }
    } else {
      const F4 y0 = ring[slot][0], y1 = ring[slot][1];
      const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xs[3] = {q < 4 ? (&x[0][0].x)[q] : (&x[0][1].x)[q - 4], q < 4 ? (&x[1][0].x)[q] : (&x[1][1].x)[q - 4], q < 4 ? (&x[2][0].x)[q] : (&x[2][1].x)[q - 4]};
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = mfma_f32_16x16x4(yv[q], xs[t], acc[t]);
        epi(q);
        sched_fence();
      }
      epi(8);
    }
    if constexpr (DBG != 1 && !(LF_EXP & 8)) gload(slot);
    sched_fence();
  };
  // A group: 8 chunks of A (row r of tile 0 at a0 + 4g, next tile 16 * kLfXs words on, chunk c at + 32 c) against NP column
  // blocks whose items alternate in the stream (chunk-major): one read of the A fragments feeds NP items.
  // (the A fragments of chunk c + 1 are requested before chunk c is multiplied: the fence in mma_item keeps that order)
  auto afrag = [&](const float* a0, int ts, int c, F4 (&x)[3][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < kLfMmaTiles; ++t) { x[t][0] = ld4(a0 + t * ts + 32 * c); x[t][1] = ld4(a0 + t * ts + 32 * c + 16); }
  };
  auto run2 = [&](const float* a0, f32x4 (&acc0)[3], f32x4 (&acc1)[3], bool pin = false) __attribute__((always_inline)) {
    F4 x[2][3][2];
    afrag(a0, 16 * kLfXs, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c + 1 < 8) afrag(a0, 16 * kLfXs, c + 1, x[(c + 1) & 1]);
      mma_item(2 * c, x[c & 1], acc0, pin);
      mma_item(2 * c + 1, x[c & 1], acc1, pin);
    }
  };
  auto run3 = [&](const float* a0, f32x4 (&acc0)[3], f32x4 (&acc1)[3], f32x4 (&acc2)[3]) __attribute__((always_inline)) {
    F4 x[2][3][2];
    afrag(a0, 16 * kLfXs, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c + 1 < 8) afrag(a0, 16 * kLfXs, c + 1, x[(c + 1) & 1]);
      mma_item(3 * c, x[c & 1], acc0);
      mma_item(3 * c + 1, x[c & 1], acc1);
      mma_item(3 * c + 2, x[c & 1], acc2);
    }
  };
  // one column block against 8 chunks (linear1 of one 128-wide hidden block)
  auto run1 = [&](const float* a0, f32x4 (&acc0)[3]) __attribute__((always_inline)) {
    F4 x[2][3][2];
    afrag(a0, 16 * kLfXs, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c + 1 < 8) afrag(a0, 16 * kLfXs, c + 1, x[(c + 1) & 1]);
      mma_item(c, x[c & 1], acc0);
    }
  };
  // both column blocks of linear2 against the 4 chunks of one hidden block (row stride kLfHs)
  auto run2h = [&](const float* a0, f32x4 (&acc0)[3], f32x4 (&acc1)[3]) __attribute__((always_inline)) {
    F4 x[2][3][2];
    afrag(a0, 16 * kLfHs, 0, x[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c + 1 < 4 && (!(LF_EXP & 4) || c == 0)) afrag(a0, 16 * kLfHs, c + 1, x[(c + 1) & 1]);
      mma_item(2 * c, x[c & 1], acc0);
      mma_item(2 * c + 1, x[c & 1], acc1);
    }
  };
  // a copy of a lane-dependent index the optimiser cannot see through: address arithmetic built on it stays where it is written
  // instead of being hoisted out of the step / layer loops and held (or spilled) across the GEMM phases
  auto opaque = [](int v) __attribute__((always_inline)) {
#if !defined(MLDHIP_SIM)
    asm volatile("" : "+v"(v));
#endif
    return v;
  };
  auto zero3 = [](f32x4 (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- this lane's elements of an operand buffer with row stride `st` words: row 16t + r, columns cw + 16 wave + 4g .. + 3 (`cw` = the
  // block's first column: 128 cb for a 256-wide buffer, 0 for the hidden block).  fp32: four consecutive words; split: two words (two
  // column pairs) in the chunk's high plane and two in its low plane (chunk = 32 words per 32 columns).
  const int swz4 = (X3 && SWZ) ? ((r >> 2) << 2) : 0;             // SWZ: XOR of a word offset inside rows 16 t + r (bits 2-3: the 16-byte group)
  const int rw0 = ((wave >> 1) * 32 + (wave & 1) * 8 + g * 2) ^ swz4;      // split image: word of the first column pair inside the 128-wide block
  const int cq0 = wave * 16 + g * 4;                              // first of this lane's four columns inside a 128-column block
  auto put_row = [&](float* buf, int st, int cw, int t, const float (&v)[4]) __attribute__((always_inline)) {
    float* row = buf + (t * 16 + r) * st + cw;
    if constexpr (X3) {
      unsigned h0, l0, h1, l1;
      split16_two(v[0], v[1], h0, l0);
      split16_two(v[2], v[3], h1, l1);
      unsigned* w = reinterpret_cast<unsigned*>(row) + rw0;
      *reinterpret_cast<U2*>(w) = U2{h0, h1};
      *reinterpret_cast<U2*>(w + 16) = U2{l0, l1};
    } else {
      st4(row + cq0, F4{v[0], v[1], v[2], v[3]});
    }
  };
  auto put = [&](float* buf, int st, int cw, const float (&val)[3][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) put_row(buf, st, cw, t, val[t]);
  };
  // ... and back: the residual of a LayerNorm is read from the operand buffer it was multiplied from (split mode: high + low
  // half, the value the GEMMs saw, 2^-22 from the fp32 one), so no activation stays in registers across a GEMM phase
  auto get = [&](const float* buf, int st, int cw, float (&val)[3][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float* row = buf + (t * 16 + r) * st + cw;
      if constexpr (X3) {
        const unsigned* w = reinterpret_cast<const unsigned*>(row) + rw0;
        const U2 h = *reinterpret_cast<const U2*>(w), l = *reinterpret_cast<const U2*>(w + 16);
        val[t][0] = f16_bits_value(h.x) + f16_bits_value(l.x);
        val[t][1] = f16_bits_value(h.x >> 16) + f16_bits_value(l.x >> 16);
        val[t][2] = f16_bits_value(h.y) + f16_bits_value(l.y);
        val[t][3] = f16_bits_value(h.y >> 16) + f16_bits_value(l.y >> 16);
      } else {
        const F4 v = ld4(row + cq0);
        val[t][0] = v.x; val[t][1] = v.y; val[t][2] = v.z; val[t][3] = v.w;
      }
    }
  };

  // LayerNorm over the 256 columns of rows 16t + r, t < nt; v[cb][t][i] = this lane's elements (columns 128 cb + 16 wave + 4g + i).
  // Part 1 publishes the per-wave row sums -- eight in-lane terms, two shuffles over g -- as red[row][wave] (the caller then passes a
  // barrier); part 2 finishes (one more barrier inside).
  auto ln_part1 = [&](const float (&v)[2][3][4], int nt) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < nt) {
        float sum = ((v[0][t][0] + v[0][t][1]) + (v[0][t][2] + v[0][t][3])) + ((v[1][t][0] + v[1][t][1]) + (v[1][t][2] + v[1][t][3]));
        sum = sum_groups(sum);
        if (g == 0) red[(t * 16 + r) * 8 + wave] = sum;
      }
    }
  };
  auto ln_part2 = [&](float (&v)[2][3][4], int nt, const float* gamma, const float* beta) __attribute__((always_inline)) {
    const F4 g0 = ld4(gamma + cq0), g1 = ld4(gamma + 128 + cq0), b0 = ld4(beta + cq0), b1 = ld4(beta + 128 + cq0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < nt) {
        const F4 ma = ld4(red + (t * 16 + r) * 8), mb = ld4(red + (t * 16 + r) * 8 + 4);
        const float mean = (((ma.x + ma.y) + (ma.z + ma.w)) + ((mb.x + mb.y) + (mb.z + mb.w))) * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[0][t][i] -= mean;
          v[1][t][i] -= mean;
          sq += v[0][t][i] * v[0][t][i] + v[1][t][i] * v[1][t][i];
        }
        sq = sum_groups(sq);
        if (g == 0) red[384 + (t * 16 + r) * 8 + wave] = sq;
      }
    }
    __syncthreads();
    const float gm0[4] = {g0.x, g0.y, g0.z, g0.w}, gm1[4] = {g1.x, g1.y, g1.z, g1.w}, bt0[4] = {b0.x, b0.y, b0.z, b0.w}, bt1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < nt) {
        const F4 qa = ld4(red + 384 + (t * 16 + r) * 8), qb = ld4(red + 384 + (t * 16 + r) * 8 + 4);
        const float rs = rsqrtf((((qa.x + qa.y) + (qa.z + qa.w)) + ((qb.x + qb.y) + (qb.z + qb.w))) * (1.0f / 256.0f) + kLnEps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[0][t][i] = v[0][t][i] * rs * gm0[i] + bt0[i];
          v[1][t][i] = v[1][t][i] * rs * gm1[i] + bt1[i];
        }
      }
    }
  };

  // token rows of one reverse step, each lane its own elements: row 16t + c, c = r; t = 0: latent + pe[0] (both CFG halves), 1: the
  // step's time row, 2: the condition rows (mld_denoiser.py:143-196; rows beyond B repeat motion B - 1, never written back)
  auto assemble = [&](int step) __attribute__((always_inline)) {
    const float* pe0 = sm_fin + 512;
    const int rq = opaque(r), cq = opaque(cq0);
    int sidx = s0 + (rq & 7);
    sidx = sidx < p.B ? sidx : p.B - 1;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const unsigned col = cb * 128 + cq;
      const F4 pe = ld4(pe0 + col), tt = ld4(p.T1 + (unsigned)step * 256u + col), la = ld4(lats + (rq & 7) * 256 + col);
      const F4 tx = ld4(p.TP + (unsigned)((rq < 8 ? 0 : p.B) + sidx) * 256u + col);
      const float xv[3][4] = {{la.x + pe.x, la.y + pe.y, la.z + pe.z, la.w + pe.w}, {tt.x, tt.y, tt.z, tt.w}, {tx.x, tx.y, tx.z, tx.w}};
      put(Xs, kLfXs, cb * 128, xv);
    }
  };

  // ---- prologue: latents; first step's token rows; the first kLoopRing items into the ring
  {
    const int c = tid >> 6, c4 = tid & 63;
    int s = s0 + c;
    s = s < p.B ? s : p.B - 1;
    const F4 v = ld4(p.init_lat + (long long)s * 256 + c4 * 4);
    st4(lats + c * 256 + c4 * 4, F4{v.x * p.init_sigma, v.y * p.init_sigma, v.z * p.init_sigma, v.w * p.init_sigma});
  }
  // ---- the small parameters of a layer (biases, LayerNorm gains: kLsLayer floats, + the 256 bias floats of the skip linear behind the
  // layer) live in LDS, double buffered: a global load of a bias in front of its use waits -- the memory counter is in order -- for every
  // weight-ring load issued before it, i.e. it drains the ring's prefetch distance once per phase (r04: the loop with all parameter reads
  // served from LDS 20.54 -> 19.92 ms at 1 280 motions).  The block of the NEXT layer is requested into 8 registers in front of the
  // out-projection's products and stored behind them: by then 32 newer ring loads are in flight, so waiting for it costs nothing.
  F4 pf0, pf1;
  auto prm_fetch = [&](int layer) __attribute__((always_inline)) {          // layer = index into the step's layer sequence (wraps to 0)
    const float* src = p.small + (unsigned)layer * (unsigned)kLsLayer;
    const int o0 = opaque(tid) * 4, o1 = 2048 + o0;
    pf0 = ld4(src + o0);
    pf1 = o1 < kLsLayer ? ld4(src + o1) : F4{0.f, 0.f, 0.f, 0.f};
    if (o1 >= kLsLayer && o1 < kLsLayer + 256) {                             // threads 320 .. 383: the skip-linear bias of this layer, if it has one
      const int si = layer - nb;
      if (si >= 0 && layer + 1 < p.L) pf1 = ld4(sm_skip + (unsigned)si * 256u + (unsigned)(o1 - kLsLayer));
    }
  };
  auto prm_store = [&](int buf) __attribute__((always_inline)) {
    float* dst = prm + buf * kLfPrmFloats;
    const int o0 = tid * 4, o1 = 2048 + o0;
    st4(dst + o0, pf0);
    if (o1 < kLfPrmFloats) st4(dst + o1, pf1);
  };
  prm_fetch(0);
  prm_store(0);
  __syncthreads();
  assemble(0);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kLoopRing; ++j) gload(j);
  int pbuf = 0;                                    // buffer of the current layer's block; toggles per layer (the layer count is odd)

  const int gs4 = X3 && SWZ ? ((g ^ (r >> 2)) << 2) : g * 4;      // this lane's 16-byte group of a half chunk (swizzled by the row)
  const float* xa = Xs + r * kLfXs + gs4;          // A fragments of the layer input
  const float* aa = As + r * kLfXs + gs4;          // ... of the attention output / the hidden-activation blocks

  if constexpr (DBG == 5) tph = clock_pinned();
  for (int step = 0; step < p.n; ++step) {
    goff = (unsigned)tid * 8u + (unsigned)(kLoopRing * kLoopItemFloats);
    for (int l = 0; l < p.L; ++l) {
      float x[2][3][4];                            // norm2 output of this layer at this lane's positions (dies inside the layer: the next one reads Xs)
      const float* sm = prm + pbuf * kLfPrmFloats;        // this layer's small parameters (LDS)
      // ================= self-attention: two heads at a time (cross_attention.py:265-266; nn.MultiheadAttention, 4 heads of 64)
      for (int hp = 0; hp < 2; ++hp) {
        f32x4 q[3], k[3], vv[3];
        zero3(q); zero3(k); zero3(vv);
        run3(xa, q, k, vv);
        stamp(0);
        const F4 bq = ld4(sm + kLsInB + hp * 128 + cq0), bk = ld4(sm + kLsInB + 256 + hp * 128 + cq0), bv = ld4(sm + kLsInB + 512 + hp * 128 + cq0);
        // partial scores of row r over this lane's four columns of the head, summed over g: the wave's 16 columns; the four waves of
        // a head meet in sc[head of the pair][(t, u)][row][wave of the head]
        const float bqv[4] = {bq.x, bq.y, bq.z, bq.w}, bkv[4] = {bk.x, bk.y, bk.z, bk.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) { q[t][i] += bqv[i]; k[t][i] += bkv[i]; vv[t][i] += bvv[i]; }
        float* sw = sc + hp * 1152 + (wave >> 2) * 576 + r * 4 + (wave & 3);      // (one exchange buffer per head pair: no barrier between the pairs)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            float sp = (q[t][0] * k[u][0] + q[t][1] * k[u][1]) + (q[t][2] * k[u][2] + q[t][3] * k[u][3]);
            sp = sum_groups(sp);
            if (g == 0) sw[(t * 3 + u) * 64] = sp;
          }
        __syncthreads();
        const float* sb = sc + hp * 1152 + (wave >> 2) * 576 + r * 4;
        float o[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float a[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const F4 e = ld4(sb + (t * 3 + u) * 64);
            a[u] = ((e.x + e.y) + (e.z + e.w)) * (0.125f * 1.44269504088896340736f);      // 1/sqrt(64), log2 domain
          }
          const float m = fmaxf(a[0], fmaxf(a[1], a[2]));
          const float e0 = fast_exp2(a[0] - m), e1 = fast_exp2(a[1] - m), e2 = fast_exp2(a[2] - m);
          const float inv = fast_rcp(e0 + e1 + e2);
          const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[t][i] = p0 * vv[0][i] + p1 * vv[1][i] + p2 * vv[2][i];
        }
        put(As, kLfXs, hp * 128, o);
        if (hp == 1) __syncthreads();       // the attention output is complete before anybody multiplies it (r04: `sc` is double buffered by head pair,
                                            // so the pairs need no barrier between them: 18.94 -> 18.83 ms at 1 280 motions)
        stamp(1);
      }
      // ================= out-projection + residual + norm1 -> Xs
      {
        f32x4 o0[3], o1[3];
        zero3(o0); zero3(o1);
        prm_fetch(l + 1 < p.L ? l + 1 : 0);
        run2(aa, o0, o1);
        prm_store(pbuf ^ 1);
        stamp(2);
        const F4 ob0 = ld4(sm + kLsOutB + cq0), ob1 = ld4(sm + kLsOutB + 128 + cq0);
        const float ob0v[4] = {ob0.x, ob0.y, ob0.z, ob0.w}, ob1v[4] = {ob1.x, ob1.y, ob1.z, ob1.w};
        float u[2][3][4];
        get(Xs, kLfXs, 0, u[0]);
        get(Xs, kLfXs, 128, u[1]);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            u[0][t][i] += o0[t][i] + ob0v[i];
            u[1][t][i] += o1[t][i] + ob1v[i];
          }
        ln_part1(u, 3);
        __syncthreads();
        ln_part2(u, 3, sm + kLsN1W, sm + kLsN1B);
        put(Xs, kLfXs, 0, u[0]);
        put(Xs, kLfXs, 128, u[1]);
        __syncthreads();
        stamp(3);
      }
      // ================= feed-forward, software pipelined over the eight 128-wide blocks of the hidden activation: the GELU /
      // split / store epilogue of block hb (VALU + LDS) is issued in the same region as linear1's MFMAs of block hb + 1 -- the two
      // are independent and run on different pipes -- then one barrier, then linear2's share of block hb accumulates in registers.
      // Two block buffers: block hb + 1 is written while a slower wave may still multiply block hb - 1 ... never the same buffer.
      {
        f32x4 y0[3], y1[3], hA[3], hB[3];
        zero3(y0); zero3(y1); zero3(hA);
        const float* ha = As + r * kLfHs + gs4;
        // block hb: epilogue of `cur` (its linear1 accumulators) next to linear1 of block hb + 1 into `nxt`, barrier, linear2's share
        auto ffn_stage = [&](int hb, f32x4 (&cur)[3], f32x4 (&nxt)[3], bool more) __attribute__((always_inline)) {
          const F4 b1 = ld4(sm + kLsL1B + hb * 128 + wave * 16 + g * 4);       // biases of this lane's four columns (transposed tile)
          float* hbuf = As + (hb & 1) * kLfHFloats;
          zero3(nxt);
          // the epilogue's three tiles leave between the matrix instructions of the next block's linear1: half a tile (two GELUs and a
          // split) per item over the first six items, in nine slices of two to five instructions (mma_item_sliced); the tile's two
          // 8-byte stores go out with the odd items
          F4 x[2][3][2];
          if (more) afrag(xa, 16 * kLfXs, 0, x[0]);
          unsigned ph = 0, pl = 0, qh = 0, ql = 0;
          float f0 = 0.f, f1 = 0.f;
          float u0 = 0.f, u1 = 0.f, hh0 = 0.f, hh1 = 0.f, tt0 = 0.f, tt1 = 0.f, ee0 = 0.f, ee1 = 0.f, pp0 = 0.f, pp1 = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            auto epi = [&](int k) __attribute__((always_inline)) {
              if (c >= 6 || DBG == 4) return;                          // DBG 4 (measurement build): no feed-forward epilogue at all
              const int t = c >> 1, i0 = (c & 1) * 2;

              constexpr bool kGelu = DBG != 3;                         // DBG 3 (measurement build): identity for GELU
              if (k == 0) {
                u0 = cur[t][i0] + (i0 ? b1.z : b1.x);
                u1 = cur[t][i0 + 1] + (i0 ? b1.w : b1.y);
                hh0 = 0.5f * u0;
                hh1 = 0.5f * u1;
              } else if (k == 1 && kGelu) {                            // rt.hpp gelu_erf, two values side by side
                tt0 = fast_rcp(fmaf(fabsf(u0), 0.3275911f * 0.70710678118654752440f, 1.0f));
                tt1 = fast_rcp(fmaf(fabsf(u1), 0.3275911f * 0.70710678118654752440f, 1.0f));
              } else if (k == 2 && kGelu) {
                const float w0 = u0 * 0.84932180028801904272f, w1 = u1 * 0.84932180028801904272f;
                ee0 = fast_exp2(-(w0 * w0));
                ee1 = fast_exp2(-(w1 * w1));
              } else if (k == 3 && kGelu) {
                pp0 = fmaf(1.061405429f, tt0, -1.453152027f);
                pp1 = fmaf(1.061405429f, tt1, -1.453152027f);
                pp0 = fmaf(pp0, tt0, 1.421413741f);
                pp1 = fmaf(pp1, tt1, 1.421413741f);
              } else if (k == 4 && kGelu) {
                pp0 = fmaf(pp0, tt0, -0.284496736f);
                pp1 = fmaf(pp1, tt1, -0.284496736f);
                pp0 = fmaf(pp0, tt0, 0.254829592f);
                pp1 = fmaf(pp1, tt1, 0.254829592f);
              } else if (k == 5 && kGelu) {
                pp0 = fmaf(-ee0, pp0 * tt0, 1.0f);
                pp1 = fmaf(-ee1, pp1 * tt1, 1.0f);
              } else if (k == 6) {
                if constexpr (kGelu) {
                  u0 = fmaf(fabsf(hh0), pp0, hh0);
                  u1 = fmaf(fabsf(hh1), pp1, hh1);
                }
                if constexpr (X3) qh = split16_hi(u0, u1);
              } else if (k == 7) {
                if constexpr (X3) {
                  ql = split16_lo(u0, u1, qh);
                  if (i0 == 0) { ph = qh; pl = ql; }
                }
              } else if (k == 8) {
                // row 16 t + r of the hidden image, columns 16 wave + 4 g .. + 3 of the 128: words (wave >> 1) 32 + (wave & 1) 8 + 2 g, + 1
                float* hrow = hbuf + (t * 16 + r) * kLfHs;
                if constexpr (X3) {
                  if (i0 != 0) {
                    unsigned* w = reinterpret_cast<unsigned*>(hrow) + rw0;
                    *reinterpret_cast<U2*>(w) = U2{ph, qh};
                    *reinterpret_cast<U2*>(w + 16) = U2{pl, ql};
                  }
                } else {
                  if (i0 == 0) { f0 = u0; f1 = u1; }
                  else st4(hrow + wave * 16 + g * 4, F4{f0, f1, u0, u1});
                }
              }
            };
            if (more) {
              if (c + 1 < 8 && (!(LF_EXP & 1) || c == 0)) afrag(xa, 16 * kLfXs, c + 1, x[(c + 1) & 1]);
              mma_item_sliced(c, x[c & 1], nxt, epi);
            } else {
#pragma unroll
              for (int k = 0; k < 9; ++k) epi(k);
            }
          }
          if (!(LF_EXP & 2)) __syncthreads();
          run2h(ha + (hb & 1) * kLfHFloats, y0, y1);
        };
        run1(xa, hA);
        for (int hq = 0; hq < 3; ++hq) {
          ffn_stage(2 * hq, hA, hB, true);
          ffn_stage(2 * hq + 1, hB, hA, true);
        }
        ffn_stage(6, hA, hB, true);
        ffn_stage(7, hB, hA, false);
        stamp(4);
        const F4 lb0 = ld4(sm + kLsL2B + cq0), lb1 = ld4(sm + kLsL2B + 128 + cq0);
        const float lb0v[4] = {lb0.x, lb0.y, lb0.z, lb0.w}, lb1v[4] = {lb1.x, lb1.y, lb1.z, lb1.w};
        get(Xs, kLfXs, 0, x[0]);
        get(Xs, kLfXs, 128, x[1]);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[0][t][i] += y0[t][i] + lb0v[i];
            x[1][t][i] += y1[t][i] + lb1v[i];
          }
        ln_part1(x, 3);
        __syncthreads();
        ln_part2(x, 3, sm + kLsN2W, sm + kLsN2B);
        stamp(7);
      }
      if (l + 1 < p.L) {
        // layer output -> Xs; first half of the stack: also parked for the skip connection (cross_attention.py:48-52)
        put(Xs, kLfXs, 0, x[0]);
        put(Xs, kLfXs, 128, x[1]);
        if (l < nb) {
          float* sk = p.skip + (size_t)(blockIdx.x * nb + l) * (48 * 256);
          const int rq = opaque(r), cq = opaque(cq0);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            float* o = sk + (unsigned)((t * 16 + rq) * 256 + cq);
            st4(o, F4{x[0][t][0], x[0][t][1], x[0][t][2], x[0][t][3]});
            st4(o + 128, F4{x[1][t][0], x[1][t][1], x[1][t][2], x[1][t][3]});
          }
        }
        __syncthreads();
        stamp(8);
        if (l >= nb) {
          // x = Linear(cat[x, skip]) (cross_attention.py:56-58): the x half of K from Xs, then the parked activation takes
          // its place in Xs for the second half
          const int si = l - nb;
          f32x4 z0[3], z1[3];
          zero3(z0); zero3(z1);
          run2(xa, z0, z1, true);
          stamp(9);
          __syncthreads();                               // everybody is done reading x
          const float* sk = p.skip + (size_t)(blockIdx.x * nb + (nb - 1 - si)) * (48 * 256);
          const int tq = opaque(tid);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int qd = tq + 512 * j, row = qd >> 6, c4 = qd & 63;
            const F4 sv = ld4(sk + (unsigned)(row * 256 + c4 * 4));
            if constexpr (X3) {
              unsigned h0, l0, h1, l1;
              split16_pair(sv.x, sv.y, h0, l0);
              split16_pair(sv.z, sv.w, h1, l1);
              unsigned* d = reinterpret_cast<unsigned*>(Xs) + row * kLfXs + (((c4 >> 3) * 32 + (c4 & 7) * 2) ^ ((X3 && SWZ) ? (((row >> 2) & 3) << 2) : 0));
              *reinterpret_cast<U2*>(d) = U2{h0, h1};
              *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
            } else {
              st4(Xs + row * kLfXs + c4 * 4, sv);
            }
          }
          __syncthreads();
          stamp(10);
          run2(xa, z0, z1, true);
          stamp(11);
          const F4 sb0 = ld4(sm + kLsLayer + cq0), sb1 = ld4(sm + kLsLayer + 128 + cq0);
          const float sb0v[4] = {sb0.x, sb0.y, sb0.z, sb0.w}, sb1v[4] = {sb1.x, sb1.y, sb1.z, sb1.w};
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              x[0][t][i] = z0[t][i] + sb0v[i];
              x[1][t][i] = z1[t][i] + sb1v[i];
            }
          __syncthreads();                               // everybody is done reading the parked rows
          put(Xs, kLfXs, 0, x[0]);
          put(Xs, kLfXs, 128, x[1]);
          __syncthreads();
        }
      } else {
      // ================= behind the last layer: end of the step: encoder.norm on the latent token (mld_denoiser.py:206), CFG (mld.py:339-342), DDIM eta = 0
      {
        ln_part1(x, 1);
        __syncthreads();
        ln_part2(x, 1, sm_fin, sm_fin + 256);
        const float sat = p.ddim[step * 4], s1mat = p.ddim[step * 4 + 1], sap = p.ddim[step * 4 + 2], s1map = p.ddim[step * 4 + 3];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          float ec[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) ec[i] = wave_xor(x[cb][0][i], 8);     // CFG row c + 8 lives in lane + 8 (same g)
          if (r < 8) {
            float* lp = lats + r * 256 + cb * 128 + cq0;
            const F4 xt = ld4(lp);
            const float xtv[4] = {xt.x, xt.y, xt.z, xt.w};
            float nv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float eu = x[cb][0][i];
              const float eps = eu + p.guidance * (ec[i] - eu);
              const float x0 = (xtv[i] - s1mat * eps) / sat;
              nv[i] = sap * x0 + s1map * eps;
            }
            st4(lp, F4{nv[0], nv[1], nv[2], nv[3]});
          }
        }
        __syncthreads();
        if (step + 1 < p.n) {
          assemble(step + 1);
          __syncthreads();
        }
        stamp(6);
      }
      }
      stamp(5);
      pbuf ^= 1;
    }
  }
  {
    const int c = tid >> 6, c4 = tid & 63;
    if (s0 + c < p.B) st4(p.lat + (long long)(s0 + c) * 256 + c4 * 4, ld4(lats + c * 256 + c4 * 4));
  }
  if constexpr (DBG == 5) {
    if (p.trace && blockIdx.x < 64 && lane == 0) {
      unsigned long long* o = p.trace + ((long long)blockIdx.x * 8 + wave) * 16;
      unsigned long long tot = 0;
      for (int k = 0; k < 15; ++k) { o[k] = ph[k]; tot += ph[k]; }
      o[15] = tot;
    }
  }
}

}  // namespace mld
