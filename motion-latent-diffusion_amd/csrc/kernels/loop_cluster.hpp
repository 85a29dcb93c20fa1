// The reverse-diffusion loop of ONE bs-64 request (up to 128 motions per launch) as one persistent launch of CLUSTERS: 24 or 12 workgroups per 8 motions.
//
// Why it exists.  The metric's literal configuration -- one MLD.forward of 64 prompts (mld.py:216-265,290-360) -- ran on
// the launch-per-GEMM family (tile32.hpp): 2 052 dependent launches of ~5.5 us, 12.3 ms per batch.  The sample-major loop (loop_fused.hpp)
// does not help there: 8 workgroups, each streaming all 30 MB of weights per step, take its flat 19 ms.  What was measured first
// (tools/loopbench/sync_bench.hip, profiles/r05_sync_bench.json): a hand-off between workgroups INSIDE a launch -- payload stores, one flag per
// producer, a relaxed poll, L1-bypassing loads -- costs 1.7 us with nothing to move and 2.7 / 3.6 / 3.1 us with 16 / 64 / 48 KB gathered per
// workgroup, against 3.8 / 4.8 / 4.3 us for the same bodies as launches.  So the loop is cut where its all-to-all points are, but the pieces stay
// in one launch:
//
//   cluster = 8 motions = 16 rows of the CFG batch x 3 tokens (the row order of loop_fused.hpp: row = 16 t + c, c < 8 unconditional).
//   member (t, h), t = token 0..2, h = column group 0..CG-1 (CG = 4: every member is a head; CG = 8, calls of up to 64 motions: members h < 4 are the
//   heads and do everything, members h >= 4 skip Ph1 and enter at E1); per layer (cross_attention.py:259-272, forward_post):
//     Ph1  [X of all 48 rows in LDS]  Q of (token t, head h) and K of head h for all three tokens on waves 0-3, V of head h for all three tokens on
//          waves 4-7 (3x redundant over t: the 3-token attention needs them), scores through LDS, softmax + P.V for its 16 rows; then the
//          out-projection SPLIT OVER K BY HEAD: the head's 64 attention dims into all 256 output columns   -> publishes a partial [16][256]
//     E1   the 4 partials of its token, summed in a fixed order + bias + residual -> norm1, one row per wave (in-wave statistics)   64 KB
//     Ph2  linear1 + GELU for hidden columns [1024 h / CG, + 1024 / CG)                      -> publishes its slice of H[16][1024] as a split-f16 image
//     E2   gather the hidden activation of its token from (t, *)                                  64 KB
//     Ph3  linear2 for output columns [256 h / CG, + 256 / CG), K = 1024 on two (CG 4) or four (CG 8) groups of waves that meet through LDS; + bias
//          + the norm2 residual from its own norm1 image                                        -> publishes Y[16][256 / CG]
//     E3   gather Y of ALL 48 rows from all members, norm2 -> next layer's X    48 KB
//   (+ per skip connection, cross_attention.py:56-58: norm2 of its own rows only, Linear(cat[x, skip]) for its 16 rows x 256 / CG columns with
//   K split like linear2's, and one more gather of all 48 rows.)  End of a step (encoder.norm, CFG, DDIM:
//   mld_denoiser.py:206, mld.py:339-346) and the next step's token rows are worked out redundantly by every member: no exchange.
//   31 exchanges per step for the 9-layer model instead of 41 launches.
//
// Weights: as in loop_fused.hpp a weight element is used by exactly one wave, so nothing is staged in LDS: `finalize` writes, per column group
// and WAVE, the fragments that wave consumes in consumption order ([fragment][lane][8 words] split-f16: high halves of k = 8g .. 8g + 7 of
// weight row 16 x + r, then the low halves) and every lane streams its 32 bytes per fragment through a 4-deep register ring (6 or 8 spill).  A head
// member streams 512 KB (CG 8) / 768 KB (CG 4) per layer; the phases that stream run at the L2 -> CU fill rate (58 B/clk), the rest of a layer is
// hand-off latency (DESIGN.md 3a has the kernel's own phase stamps).
//
// Hand-offs: cdna_hip_programming.md Guideline 16 form R1 -- write-through (sc1) payload stores, every storing wave drains its memory counter,
// barrier, one lane stores the member's flag (relaxed, agent scope; value = epoch, never reset inside a call; zeroed by clear_cluster_flags_kernel in
// front of the launch -- NOT by a memset node: DESIGN.md 3a), consumers poll the flags of the producers they need with one relaxed load per lane, then
// read with sc1 loads.  PLAIN payload stores are 0.15-0.6 us per exchange cheaper but only visible to consumers behind the SAME L2: WT = false is
// used only when every member of the cluster reports the same HW_REG_XCC_ID (checked in the kernel's first exchange; block b runs on XCD b % 8 in
// practice, not by contract); a cluster that spans XCDs keeps the write-through stores by itself.
// Buffers are double buffered by the parity of the epoch; E3-type waits cover ALL members even where fewer rows are read, which is what
// keeps a fast member from overwriting a buffer a slow one still reads.  Every spin is bounded: a member that waits longer than
// kClTimeoutTicks sets the call's status words, every member that sees it leaves, the latents are poisoned with NaN (counted by the range
// contract's non-finite counter) and the handle leaves the cluster loop (mldhip_numeric_status).  The last member to finish clears the cluster's flag
// line, so every polled word is zero when a call ends.  The launch needs all its workgroups resident together: the engine sizes it to the chip and never
// issues two of them side by side (engine/params.hpp ClusterLane).
#pragma once
#include "loop_fused.hpp"

namespace mld {

#ifndef CL_RING
#define CL_RING 4
#endif
constexpr int kClMembers = 12, kClMembersMax = 24, kClRing = CL_RING, kClFragFloats = 512;   // members: 3 tokens x 4 column groups (x 8: the wide form)
constexpr int kClXs = 264, kClHs = 1032;                       // LDS row strides (words), = 8 mod 16: conflict-free fragment reads
constexpr unsigned kClPlane = 2u * 48u * 256u;                 // one double-buffered [48][256] fp32 exchange tensor (floats)
constexpr unsigned kClSlab = 2u * 12u * 16u * 256u;            // one double-buffered set of per-member partial slabs [3 tokens][4 members][16][256] (floats)
constexpr unsigned kClH1 = 0, kClY = kClPlane, kClZ = 2 * kClPlane, kClH = 3 * kClPlane, kClPO = kClH + 2u * 48u * 1024u;
constexpr unsigned kClXFloats = kClPO + kClSlab;               // exchange region of one cluster: 270 336 floats = 1 056 KB
constexpr int kClAoS = 72;                                     // LDS row stride (words) of the head's attention output image [16][64 columns]
constexpr int kClFlagLine = 32, kClFlagWords = 4 * kClFlagLine; // flag kinds AO, H, Y, Z: 32 words (one 128-byte line) each; word 28 of the Z line counts finished members
enum : int { kFlagAO = 0, kFlagH = 1, kFlagY = 2, kFlagZ = 3 };
constexpr int kClBigFloats = 16 * kClHs;                       // X of all 48 rows ([48][264] = 12 672 words) and the token's hidden activation ([16][1032]) in turn
static_assert(kClBigFloats >= 48 * kClXs, "X and the hidden activation share one region");
constexpr int kClAsFloats = 16 * kClXs;
constexpr int kClScFloats = 3 * 16 * 4, kClRedFloats = 2 * 16 * 8, kClRed2Floats = 6 * 64 * 4, kClCtlFloats = 16;
constexpr int kClLdsFloats = kClBigFloats + kClAsFloats + kLfLatFloats + 2 * kLfPrmFloats + kClScFloats + kClRedFloats + kClRed2Floats + kClCtlFloats;
constexpr int kClLdsBytes = kClLdsFloats * 4;                  // 125 760 B: one workgroup per CU
constexpr int kClMaxClusters = 16;                             // two clusters per XCD (32 CUs): 128 motions per launch
constexpr int kClMaxCall = 2 * 8 * kClMaxClusters;             // motions per call the engine serves with (two) cluster launches

struct ClFrag { long long src; int ld; int pad; };            // element [row0][k0] of a weight (floats into the arena), row stride

struct ClusterArgs {
  const float* stream;        // per column group and wave: [fragments of a step + kClRing][64 lanes][8 words]
  const unsigned* wave_off;   // [column group][wave] (32 or 64 words in device memory): float offset of that wave's fragment sequence
  const float* small;         // loop_fused.hpp's packed small parameters (kLs*)
  const float* T1;            // [n][256] time-token rows
  const float* TP;            // [2B][256] condition-token rows, unconditional half first
  const float* init_lat;      // [B][256]
  float* lat;                 // [B][256]
  float* park;                // [workgroup][nb][16][256] parked skip rows of the member's own token
  const float* ddim;          // [n][4]
  float* xbuf;                // [clusters][kClXFloats] exchange regions
  unsigned* flags;            // [clusters][kClFlagWords], zeroed in front of the launch
  unsigned* status;           // [0]: 0 ok, 1 a wait timed out; [1]: clusters that span XCDs (plain stores asked for, write-through used); both cleared per launch.  [2]: a wait timed out in SOME launch since the host last looked (sticky)
  int B, L, n, ncl;           // B: motions of the CALL (the condition rows' pitch); this launch serves motions [s_base, s_end)
  int s_base = 0, s_end = 0;
  unsigned* host_status = nullptr;   // pinned host word (or NULL): set with the sticky status word, read by the host at the start of the next call without a device synchronisation
  unsigned timeout = 0;       // wait bound: 100 MHz ticks (GPU) / poll iterations (simulator); kClTimeoutTicks unless a test shortens it
  int mute = -1;              // hooks builds (fault injection, tests): this member never raises its first flag -- everybody who waits for it runs into the bound
  unsigned long long* trace = nullptr;   // CL_TRACE builds (tools/loopbench only): [workgroup][wave][16] shader cycles per phase, summed over steps and layers
  int xslots;                 // blocks per launch row: 8 on the GPU (block b runs on XCD b % 8: a cluster's members share the slot), min(clusters, 8) on the simulator
  float guidance, init_sigma;
};

#if defined(MLDHIP_SIM)
constexpr unsigned kClTimeoutTicks = 1u << 26;                 // poll iterations
#else
constexpr unsigned kClTimeoutTicks = 20000000u;                // 100 MHz ticks: 200 ms
#endif

// finalize-time: one fragment = 16 weight rows x 32 k as the split-f16 operand pair of lane (r, g): high halves of W[row0 + r][k0 + 8g .. + 7], then the low halves
__global__ __launch_bounds__(64) void pack_cluster_frags_kernel(const float* __restrict__ arena, const ClFrag* __restrict__ frags, float* __restrict__ out) {
  const ClFrag f = frags[blockIdx.x];
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const float* src = arena + f.src + (long long)r * f.ld + 8 * g;
  const F4 a = ld4(src), b = ld4(src + 4);
  U4 hi, lo;
  split16_pair(a.x, a.y, hi.x, lo.x);
  split16_pair(a.z, a.w, hi.y, lo.y);
  split16_pair(b.x, b.y, hi.z, lo.z);
  split16_pair(b.z, b.w, hi.w, lo.w);
  float* dst = out + (long long)blockIdx.x * kClFragFloats + lane * 8;
  *reinterpret_cast<U4*>(dst) = hi;
  *reinterpret_cast<U4*>(dst + 4) = lo;
}

// every polled word zero at the start of a call (Guideline 16 "Re-initialise every call"): a kernel of the call's own stream, not a memset node
__global__ __launch_bounds__(256) void clear_cluster_flags_kernel(unsigned* __restrict__ flags, int words) {
  for (int i = threadIdx.x; i < words; i += 256) flag_store(flags + i, 0u);
}

// hooks builds (option "cluster_stale"): one polled word holds an epoch no fresh launch can hold -- the entry check's test
__global__ void poke_cluster_flag_kernel(unsigned* __restrict__ word, unsigned v) { flag_store(word, v); }

// grid = 12 xslots x ceil(clusters / xslots), xslots = 8: block b -> XCD slot x = b % 8, index i = b / 8 -> cluster x + 8 (i / 12), member i % 12.  block = 512.
// WT = true: write-through (sc1) payload stores whatever the placement.  WT = false: every cluster whose twelve members report the same XCC id stores its
// payloads plain (served from the shared L2); a cluster that spans XCDs falls back to write-through by itself.
// CG = column groups per token: 4 (12 members: every member is a head) or 8 (24 members, calls of up to 8 clusters: the feed-forward block on twice the CUs --
// members (t, h < 4) are the heads and do everything, members (t, h >= 4) skip Ph1 and enter at E1; linear1 = 128 hidden columns per member, linear2 and the
// skip linear = 32 output columns per member with K on four wave pairs).
template <bool WT, int CG>
__global__ __launch_bounds__(512, 2) void den_cluster_kernel(ClusterArgs p) {
  constexpr int kM = 3 * CG;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* Xs = smem;                              // [48][264] layer input, split image; later [16][1032]: the token's hidden activation
  float* As = Xs + kClBigFloats;                 // [16][264] attention output, then norm1 output (own token), split image
  float* lats = As + kClAsFloats;                // [8][256]
  float* prm = lats + kLfLatFloats;              // [2][kLfPrmFloats]
  float* sc = prm + 2 * kLfPrmFloats;            // [3 keys][16 rows][4 waves] partial attention scores
  float* red = sc + kClScFloats;                 // [2 passes][16 rows][8 waves]
  float* red2 = red + kClRedFloats;              // [4 waves][64 lanes][4] K-half partial tiles
  unsigned* ctl = reinterpret_cast<unsigned*>(red2 + kClRed2Floats);
  const int wave = wave_uniform((int)threadIdx.x >> 6);
  unsigned goff = 0;                                       // this lane's running word offset into its wave's fragment stream
  [[maybe_unused]] const int hc_ = (((int)blockIdx.x / p.xslots) % kM) % CG;
  int lane = (int)threadIdx.x & 63, tid = (int)threadIdx.x, r = lane & 15, g = lane >> 4;
  int swz4 = ((r >> 2) & 3) << 2;                          // row swizzle of the operand images (loop_fused.hpp SWZ): XOR of the word offset's bits 2-3
  int gs4 = (g << 2) ^ swz4;                               // this lane's 16-byte group of a half chunk
  // Lane-dependent indices are laundered at the top of every phase: address arithmetic built on them is then redone where it is used instead of
  // being hoisted out of the step / layer loops and held -- i.e. spilled -- across all phases (loop_fused.hpp `opaque`; the first build of this kernel:
  // 256 registers + 544 B of scratch per lane, most of it loop-invariant addresses stored in the prologue)
  auto fresh = [&]() __attribute__((always_inline)) {
#if !defined(MLDHIP_SIM)
    asm volatile("" : "+v"(lane));
#endif
#if defined(CL_EXP) && (CL_EXP & 1)
    goff = p.wave_off[hc_ * 8 + wave] + (unsigned)lane * 8u;      // measurement build (WRONG results, tools/loopbench only): every phase re-reads the step's first fragments -- an L2-resident weight stream
#endif
    tid = wave * 64 + lane;
    r = lane & 15;
    g = lane >> 4;
    swz4 = ((r >> 2) & 3) << 2;
    gs4 = (g << 2) ^ swz4;
  };
  const int bx = (int)blockIdx.x % p.xslots, bi = (int)blockIdx.x / p.xslots;
  const int cluster = bx + p.xslots * (bi / kM), member = bi % kM;
  if (cluster >= p.ncl) return;
  const int tk = member / CG, hc = member % CG;  // token, column group (a head when < 4)
  const bool att = hc < 4;
  constexpr unsigned all_mask = (1u << kM) - 1u;
  const int s0 = p.s_base + cluster * 8, nb = (p.L - 1) / 2;
  const float* sm_fin = p.small + (long long)p.L * kLsLayer + nb * 256;
  const XBuf xb = xbuf_make(p.xbuf + (size_t)cluster * kClXFloats, kClXFloats * 4u);
  unsigned* flags = p.flags + (size_t)cluster * kClFlagWords;

#ifdef CL_TRACE
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tph = clock_light();
#define CL_STAMP(k) do { const unsigned long long t_ = clock_light(); ph[k] += t_ - tph; tph = t_; } while (0)
#else
#define CL_STAMP(k) do { } while (0)
#endif
  // ---- waiting for members: wave 0, lane m polls member m's flag of `kind`; bounded; the verdict reaches everybody through LDS + barrier
  auto wait_flags = [&](int kind, unsigned mask, unsigned epoch) -> bool {
    if (wave == 0) {
      const unsigned* f = flags + kind * kClFlagLine + (lane < kM ? lane : 0);
      const bool need = lane < kM && ((mask >> lane) & 1u);
      bool ok = true;
#if defined(MLDHIP_SIM)
      unsigned long long it = 0;
#else
      const unsigned long long ts = realtime_100mhz();
      unsigned it = 0;
#endif
      for (;;) {
        const bool ready = !need || flag_load(f) >= epoch;
        if (!wave_any(!ready)) break;
        spin_pause();
#if defined(MLDHIP_SIM)
        if (wave_any(++it > p.timeout || flag_load(p.status) != 0u)) { ok = false; break; }      // (wave-uniform exit: the lanes meet again in wave_any)
#else
        if ((++it & 63u) == 0u) {
          const bool late = realtime_100mhz() - ts > p.timeout;
          if (wave_any(late || flag_load(p.status) != 0u)) { ok = false; break; }
        }
#endif
      }
      poll_fence();
      if (lane == 0) {
        if (!ok) { flag_store(p.status, 1u); flag_store(p.status + 2, 1u); host_flag_store(p.host_status, 1u); }      // [2] is sticky: cleared by the host once it has acted on it
        ctl[0] = ok ? 1u : 0u;
      }
    }
    __syncthreads();
    return ctl[0] != 0u;
  };
  // ---- one wave waits for ONE member (the producer of the slice this wave gathers): no workgroup barrier between the poll and the loads; a timeout is left in ctl[2] for
  // everybody to see behind the next barrier (12-workgroup form's E2: -1.6 % per layer; on the 24-workgroup form eight waves polling eight words of one line lose 1.5 %)
  auto wait_one = [&](int kind, int m, unsigned epoch) {
    const unsigned* f = flags + kind * kClFlagLine + m;
    bool ok = true;
#if defined(MLDHIP_SIM)
    unsigned long long it = 0;
#else
    const unsigned long long ts = realtime_100mhz();
    unsigned it = 0;
#endif
    for (;;) {
      if (!wave_any(flag_load(f) < epoch)) break;
      spin_pause();
#if defined(MLDHIP_SIM)
      if (wave_any(++it > p.timeout || flag_load(p.status) != 0u)) { ok = false; break; }
#else
      if ((++it & 63u) == 0u) {
        const bool late = realtime_100mhz() - ts > p.timeout;
        if (wave_any(late || flag_load(p.status) != 0u)) { ok = false; break; }
      }
#endif
    }
    poll_fence();
    if (!ok && lane == 0) { flag_store(p.status, 1u); flag_store(p.status + 2, 1u); host_flag_store(p.host_status, 1u); ctl[2] = 1u; }
  };
  auto publish = [&](int kind, unsigned epoch) __attribute__((always_inline)) {
#if defined(CL_EXP) && (CL_EXP & 2)
    // measurement build (WRONG results, tools/loopbench only): the flag goes up without waiting for the payload stores or for the other waves -- the upper bound of what
    // cheaper publishes (per-wave flags, no workgroup barrier) could buy (profiles/r06_loop_experiments.json)
    if (tid == 0) flag_store(flags + kind * kClFlagLine + member, epoch);
    return;
#endif
    drain_stores();
    __syncthreads();
    if (tid == 0 && !(member == p.mute && epoch == 1u)) flag_store(flags + kind * kClFlagLine + member, epoch);
  };
  auto give_up = [&]() {           // a wait failed: poison this cluster's latents (member 0), leave
    if (member == 0) {
      const int c = tid >> 6, c4 = tid & 63;
      const float qnan = __builtin_nanf("");
      if (s0 + c < p.s_end) st4(p.lat + (long long)(s0 + c) * 256 + c4 * 4, F4{qnan, qnan, qnan, qnan});
    }
  };

  // Entry check (advisor r5): what a FRESH launch can find in this cluster's polled words is bounded -- a member is at most one exchange ahead of member 0 (it needs
  // member 0's share to go on), so the AO / H / Y lines hold epochs <= 2, the Z line the XCC census (<= 16) and its finish counter 0.  Anything else is what a previous
  // launch left behind and the clear in front of this one did not remove (r05: a captured memset node): the launch is failed (status words, NaN latents, counted, fallback;
  // the other members see the status word in their waits) instead of consuming the words as "ready".
  if (member == 0) {
    if (wave == 0) {
      const unsigned a = flag_load(flags + lane), b = flag_load(flags + 64 + lane);
      const unsigned lim = lane < 32 ? 2u : (lane == 32 + 28 ? 0u : 17u);
      const bool stale = wave_any(a > 2u || b > lim);
      if (lane == 0) {
        ctl[0] = stale ? 0u : 1u;
        if (stale) { flag_store(p.status, 1u); flag_store(p.status + 2, 1u); host_flag_store(p.host_status, 1u); }
      }
    }
    __syncthreads();
    if (ctl[0] == 0u) { give_up(); return; }
  }

  // ---- weight ring: this lane's two MFMA operands (32 bytes) of the wave's next kClRing fragments
  const unsigned wbase = p.wave_off[hc * 8 + wave] + (unsigned)lane * 8u;
  goff = wbase;
  F4 ring[kClRing][2];
  auto gload = [&](int slot) __attribute__((always_inline)) {
    const float* s = p.stream + goff;
    ring[slot][0] = ld4(s);
    ring[slot][1] = ld4(s + 4);
    goff += (unsigned)kClFragFloats;
#if !defined(MLDHIP_SIM)
    asm volatile("" : "+v"(goff));
#endif
  };
  // three tokens against one fragment (loop_fused.hpp mma_item); one token against one fragment on two accumulators (cross terms / high x high)
  // (`more` = false for the last kClRing fragments in front of a publish: the ring is refilled BEHIND the flag store, under the wait -- a drain of the
  // payload stores would otherwise wait for the look-ahead loads issued just before them: the memory counter is in order)
  auto mma3 = [&](int j, const F4 (&x)[3][2], f32x4 (&acc)[3], bool more = true) __attribute__((always_inline)) {
    const int slot = j % kClRing;
    const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][1]), acc[t]);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = mfma_x3_16x16x32(wl, __builtin_bit_cast(U4, x[t][0]), acc[t]);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[t][0]), acc[t]);
    if (more) gload(slot);
    sched_fence();
  };
  auto mma1 = [&](int j, const F4 (&x)[2], f32x4& a0, f32x4& a1, bool more = true) __attribute__((always_inline)) {
    const int slot = j % kClRing;
    const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
    a0 = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[1]), a0);
    a1 = mfma_x3_16x16x32(wh, __builtin_bit_cast(U4, x[0]), a1);
    a0 = mfma_x3_16x16x32(wl, __builtin_bit_cast(U4, x[0]), a0);
    if (more) gload(slot);
    sched_fence();
  };
  auto refill = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < kClRing; ++j) gload(j);
  };
  auto frag = [&](const float* buf, int st, int row, int kc, F4 (&x)[2]) __attribute__((always_inline)) {
    const float* a = buf + row * st + 32 * kc + gs4;
    x[0] = ld4(a);
    x[1] = ld4(a + 16);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  bool wt_ = true;                                         // payload stores write-through (set after the placement census)
  auto xst4 = [&](unsigned off, F4 v) __attribute__((always_inline)) { if (wt_) xbuf_st4<true>(xb, off, v); else xbuf_st4<false>(xb, off, v); };
  auto xst2 = [&](unsigned off, U2 v) __attribute__((always_inline)) { if (wt_) xbuf_st2<true>(xb, off, v); else xbuf_st2<false>(xb, off, v); };

  // ---- row-per-wave helpers (gathers, LayerNorm over whole rows, token assembly): lane l owns columns 4l .. 4l + 3 of a row
  auto st_row = [&](float* buf, int st, int row, F4 v) __attribute__((always_inline)) {       // -> split image, swizzled by the row
    unsigned h0, l0, h1, l1;
    split16_two(v.x, v.y, h0, l0);
    split16_two(v.z, v.w, h1, l1);
    unsigned* d = reinterpret_cast<unsigned*>(buf) + row * st + ((((lane >> 3) << 5) + ((lane & 7) << 1)) ^ (((row >> 2) & 3) << 2));
    *reinterpret_cast<U2*>(d) = U2{h0, h1};
    *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
  };
  auto ln_rows = [&](F4 (&v)[6], int nr, const float* gamma, const float* beta) __attribute__((always_inline)) {
    const F4 gm = ld4(gamma + lane * 4), bt = ld4(beta + lane * 4);
    float s[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = i < nr ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
#if defined(CL_EXP) && (CL_EXP & 4)
    // measurement build (WRONG results, tools/loopbench only): no cross-lane reductions in the LayerNorms -- the upper bound of what row statistics published by the
    // producers of Y / the out-projection partials could buy
#define CL_SUM64(x) (x)
#else
#define CL_SUM64(x) sum64(x)
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i < nr) s[i] = CL_SUM64(s[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < nr) {
        const float mean = s[i] * (1.0f / 256.0f);
        v[i] = F4{v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean};
        s[i] = (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i < nr) s[i] = CL_SUM64(s[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < nr) {
        const float rs = rsqrtf(s[i] * (1.0f / 256.0f) + kLnEps);
        v[i] = F4{v[i].x * rs * gm.x + bt.x, v[i].y * rs * gm.y + bt.y, v[i].z * rs * gm.z + bt.z, v[i].w * rs * gm.w + bt.w};
      }
    }
  };
  auto f4add = [](F4 a, F4 b) { return F4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; };

  // token rows of a reverse step -> Xs (mld_denoiser.py:143-196): this wave's rows w + 8 i: token 0 = latent + pe[0] (both CFG halves of motion w),
  // token 1 = the step's time row, token 2 = the condition rows (unconditional, conditional)
  auto assemble = [&](int step) __attribute__((always_inline)) {
    const float* pe0 = sm_fin + 512;
    int sidx = s0 + wave;
    sidx = sidx < p.s_end ? sidx : p.s_end - 1;
    const F4 pe = ld4(pe0 + lane * 4), la = ld4(lats + wave * 256 + lane * 4), tt = ld4(p.T1 + (unsigned)step * 256u + lane * 4);
    const F4 tu = ld4(p.TP + (unsigned)sidx * 256u + lane * 4), tc = ld4(p.TP + (unsigned)(p.B + sidx) * 256u + lane * 4);
    const F4 x0 = f4add(la, pe);
    st_row(Xs, kClXs, wave, x0);
    st_row(Xs, kClXs, wave + 8, x0);
    st_row(Xs, kClXs, 16 + wave, tt);
    st_row(Xs, kClXs, 24 + wave, tt);
    st_row(Xs, kClXs, 32 + wave, tu);
    st_row(Xs, kClXs, 40 + wave, tc);
  };
  // a layer's small parameters (+ the bias of the skip linear behind it) -> LDS, double buffered (loop_fused.hpp prm_fetch / prm_store)
  F4 pf0, pf1;
  auto prm_fetch = [&](int layer) __attribute__((always_inline)) {
    const float* src = p.small + (unsigned)layer * (unsigned)kLsLayer;
    const int o0 = tid * 4, o1 = 2048 + o0;
    pf0 = ld4(src + o0);
    pf1 = o1 < kLsLayer ? ld4(src + o1) : F4{0.f, 0.f, 0.f, 0.f};
    if (o1 >= kLsLayer && o1 < kLsLayer + 256) {
      const int si = layer - nb;
      if (si >= 0 && layer + 1 < p.L) pf1 = ld4(p.small + (unsigned)p.L * (unsigned)kLsLayer + (unsigned)si * 256u + (unsigned)(o1 - kLsLayer));
    }
  };
  auto prm_store = [&](int buf) __attribute__((always_inline)) {
    float* dst = prm + buf * kLfPrmFloats;
    const int o0 = tid * 4, o1 = 2048 + o0;
    st4(dst + o0, pf0);
    if (o1 < kLfPrmFloats) st4(dst + o1, pf1);
  };

  // ---- prologue: latents, parameters of layer 0, the first step's rows, the ring; placement census when plain stores were asked for
  {
    const int c = tid >> 6, c4 = tid & 63;
    int s = s0 + c;
    s = s < p.s_end ? s : p.s_end - 1;
    const F4 v = ld4(p.init_lat + (long long)s * 256 + c4 * 4);
    st4(lats + c * 256 + c4 * 4, F4{v.x * p.init_sigma, v.y * p.init_sigma, v.z * p.init_sigma, v.w * p.init_sigma});
  }
  prm_fetch(0);
  prm_store(0);
  if constexpr (CG == 4) { if (tid == 0) ctl[2] = 0u; }      // (wait_one of the 12-workgroup form)
  __syncthreads();
  assemble(0);
#pragma unroll
  for (int j = 0; j < kClRing; ++j) gload(j);
  bool wt = true;
  if constexpr (!WT) {
    // every member posts 1 + its XCC id as its Z flag (Z epochs start above 16: see below); a cluster that spans XCDs keeps the write-through stores
    if (tid == 0) flag_store(flags + kFlagZ * kClFlagLine + member, 1u + xcc_id());
    if (!wait_flags(kFlagZ, all_mask, 1u)) { give_up(); return; }
    if (wave == 0) {
      const unsigned mine = 1u + xcc_id();
      const unsigned other = lane < kM ? flag_load(flags + kFlagZ * kClFlagLine + lane) : mine;
      const bool spans = wave_any(other != mine);
      if (lane == 0) { ctl[1] = spans ? 1u : 0u; if (spans && member == 0) flag_store(p.status + 1, 1u); }
    }
    __syncthreads();
    wt = ctl[1] != 0u;
  }
  wt_ = wt;
  __syncthreads();
  int pbuf = 0;
  const unsigned wg = blockIdx.x;
  // linear2 output + bias + residual (both added by its producer) of row `row`, this lane's 4 columns
  auto y_row = [&](unsigned par, int row) __attribute__((always_inline)) {
    return xbuf_ld4(xb, (kClY + par * 12288u + (unsigned)(row * 256 + lane * 4)) * 4u);
  };
  // norm1 output of the own token, row r, columns c0 .. c0 + 3 as the GEMMs saw it (high + low half of the As image): the residual of norm2
  auto h1_res = [&](int c0) __attribute__((always_inline)) {
    const int l4 = c0 >> 2;
    const unsigned* wq = reinterpret_cast<const unsigned*>(As) + r * kClXs + ((((l4 >> 3) << 5) + ((l4 & 7) << 1)) ^ swz4);
    const U2 h = *reinterpret_cast<const U2*>(wq), lo = *reinterpret_cast<const U2*>(wq + 16);
    return F4{f16_bits_value(h.x) + f16_bits_value(lo.x), f16_bits_value(h.x >> 16) + f16_bits_value(lo.x >> 16),
              f16_bits_value(h.y) + f16_bits_value(lo.y), f16_bits_value(h.y >> 16) + f16_bits_value(lo.y >> 16)};
  };

  for (int step = 0; step < p.n; ++step) {
    goff = wbase + (unsigned)(kClRing * kClFragFloats);
    for (int l = 0; l < p.L; ++l) {
      const float* sm = prm + pbuf * kLfPrmFloats;
      const unsigned epoch = (unsigned)(step * p.L + l) + 1u;
      const unsigned par = epoch & 1u;
      const unsigned own_mask = ((1u << CG) - 1u) << (CG * tk), ao_mask = 0xFu << (CG * tk);
      fresh();
      // ================= Ph1: Q (own token), K, V (all tokens) of head hc; 3-token attention for the 16 rows of token tk
      if (att) {
      if (wave < 4) {
        f32x4 q0 = zero4, q1 = zero4, k[3] = {zero4, zero4, zero4};
        F4 x[2][3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) frag(Xs, kClXs, 16 * t + r, 0, x[0][t]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 3; ++t) frag(Xs, kClXs, 16 * t + r, kc + 1, x[(kc + 1) & 1][t]);
          }
          // the own token's fragments, selected without a dynamic register index
          F4 xo[2];
          xo[0] = tk == 0 ? x[kc & 1][0][0] : (tk == 1 ? x[kc & 1][1][0] : x[kc & 1][2][0]);
          xo[1] = tk == 0 ? x[kc & 1][0][1] : (tk == 1 ? x[kc & 1][1][1] : x[kc & 1][2][1]);
          mma1(2 * kc, xo, q0, q1);
          mma3(2 * kc + 1, x[kc & 1], k);
        }
        const F4 bq = ld4(sm + kLsInB + hc * 64 + wave * 16 + g * 4), bk = ld4(sm + kLsInB + 256 + hc * 64 + wave * 16 + g * 4);
        const float qv[4] = {q0[0] + q1[0] + bq.x, q0[1] + q1[1] + bq.y, q0[2] + q1[2] + bq.z, q0[3] + q1[3] + bq.w};
        const float bkv[4] = {bk.x, bk.y, bk.z, bk.w};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          float sp = (qv[0] * (k[u][0] + bkv[0]) + qv[1] * (k[u][1] + bkv[1])) + (qv[2] * (k[u][2] + bkv[2]) + qv[3] * (k[u][3] + bkv[3]));
          sp = sum_groups(sp);
          if (g == 0) sc[(u * 16 + r) * 4 + wave] = sp;
        }
        __syncthreads();
      } else {
        const int w4 = wave - 4;
        f32x4 v[3] = {zero4, zero4, zero4};
        F4 x[2][3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) frag(Xs, kClXs, 16 * t + r, 0, x[0][t]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 3; ++t) frag(Xs, kClXs, 16 * t + r, kc + 1, x[(kc + 1) & 1][t]);
          }
          mma3(kc, x[kc & 1], v);
        }
        const F4 bv = ld4(sm + kLsInB + 512 + hc * 64 + w4 * 16 + g * 4);
        const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
        __syncthreads();
        float a[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const F4 e = ld4(sc + (u * 16 + r) * 4);
          a[u] = ((e.x + e.y) + (e.z + e.w)) * (0.125f * 1.44269504088896340736f);      // 1 / sqrt(64), log2 domain
        }
        const float m = fmaxf(a[0], fmaxf(a[1], a[2]));
        const float e0 = fast_exp2(a[0] - m), e1 = fast_exp2(a[1] - m), e2 = fast_exp2(a[2] - m);
        const float inv = fast_rcp(e0 + e1 + e2);
        const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (p0 * (v[0][i] + bvv[i]) + p1 * (v[1][i] + bvv[i])) + p2 * (v[2][i] + bvv[i]);
        // attention output of head hc, row r, head dims 16 w4 + 4 g .. + 3 -> split image [16][64] in the As region (chunk w4 >> 1 of two)
        unsigned h0, l0, h1, l1;
        split16_two(o[0], o[1], h0, l0);
        split16_two(o[2], o[3], h1, l1);
        unsigned* wq = reinterpret_cast<unsigned*>(As) + r * kClAoS + (w4 >> 1) * 32 + (((w4 & 1) * 8 + g * 2) ^ swz4);
        *reinterpret_cast<U2*>(wq) = U2{h0, h1};
        *reinterpret_cast<U2*>(wq + 16) = U2{l0, l1};
      }
      __syncthreads();
      {
        // out-projection, split over K by HEAD: this member multiplies its head's 64 attention dims into all 256 output columns (this wave: 32 w .. + 31) and
        // publishes the raw partial; the four partials of a token are summed by everybody in E1 (one quarter of the weight bytes of a full out-projection per member)
        f32x4 a[2][2] = {{zero4, zero4}, {zero4, zero4}};
        F4 x[2][2];
        frag(As, kClAoS, r, 0, x[0]);
        frag(As, kClAoS, r, 1, x[1]);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          mma1(2 * kc, x[kc], a[0][0], a[0][1], false);      // (16 or 8 fragments consumed so far in this phase: the ring slot is the same)
          mma1(2 * kc + 1, x[kc], a[1][0], a[1][1], false);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 pj = a[j][0] + a[j][1];
          xst4((kClPO + par * 49152u + (unsigned)(((tk * 4 + hc) * 16 + r) * 256 + wave * 32 + j * 16 + g * 4)) * 4u, F4{pj[0], pj[1], pj[2], pj[3]});
        }
      }
      CL_STAMP(0);
      publish(kFlagAO, epoch);
      refill();
      }
      CL_STAMP(1);
      fresh();
      // ================= E1: attention output of the token from its four members -> As
      if (!wait_flags(kFlagAO, ao_mask, epoch)) { give_up(); return; }
      CL_STAMP(2);
      {
        // rows w and w + 8 of the token: sum of the four partials (fixed order) + bias + residual (the layer input as the GEMMs saw it) -> norm1, in-wave
        F4 v[6];
        const F4 ob = ld4(sm + kLsOutB + lane * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wave + 8 * i;
          F4 acc4 = xbuf_ld4(xb, (kClPO + par * 49152u + (unsigned)(((tk * 4 + 0) * 16 + row) * 256 + lane * 4)) * 4u);
#pragma unroll
          for (int m = 1; m < 4; ++m) acc4 = f4add(acc4, xbuf_ld4(xb, (kClPO + par * 49152u + (unsigned)(((tk * 4 + m) * 16 + row) * 256 + lane * 4)) * 4u));
          const unsigned* wq = reinterpret_cast<const unsigned*>(Xs) + (16 * tk + row) * kClXs + ((((lane >> 3) << 5) + ((lane & 7) << 1)) ^ (((row >> 2) & 3) << 2));
          const U2 h = *reinterpret_cast<const U2*>(wq), lo = *reinterpret_cast<const U2*>(wq + 16);
          v[i] = F4{acc4.x + ob.x + (f16_bits_value(h.x) + f16_bits_value(lo.x)), acc4.y + ob.y + (f16_bits_value(h.x >> 16) + f16_bits_value(lo.x >> 16)),
                    acc4.z + ob.z + (f16_bits_value(h.y) + f16_bits_value(lo.y)), acc4.w + ob.w + (f16_bits_value(h.y >> 16) + f16_bits_value(lo.y >> 16))};
        }
        CL_STAMP(3);
        ln_rows(v, 2, sm + kLsN1W, sm + kLsN1B);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wave + 8 * i;
          unsigned h0, l0, h1, l1;
          split16_two(v[i].x, v[i].y, h0, l0);
          split16_two(v[i].z, v[i].w, h1, l1);
          unsigned* d = reinterpret_cast<unsigned*>(As) + row * kClXs + ((((lane >> 3) << 5) + ((lane & 7) << 1)) ^ (((row >> 2) & 3) << 2));
          *reinterpret_cast<U2*>(d) = U2{h0, h1};
          *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
        }
      }
      __syncthreads();
      {
        F4 x[2][2];
        CL_STAMP(4);
        fresh();
        // linear1 + GELU
        if constexpr (CG == 4) {
        f32x4 h[2][2] = {{zero4, zero4}, {zero4, zero4}};
        frag(As, kClXs, r, 0, x[0]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) frag(As, kClXs, r, kc + 1, x[(kc + 1) & 1]);
          mma1(2 * kc, x[kc & 1], h[0][0], h[0][1], 2 * kc + kClRing < 16);
          mma1(2 * kc + 1, x[kc & 1], h[1][0], h[1][1], 2 * kc + 1 + kClRing < 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const F4 b1 = ld4(sm + kLsL1B + hc * 256 + wave * 32 + j * 16 + g * 4);
          const float v0 = gelu_erf((h[j][0][0] + h[j][1][0]) + b1.x), v1 = gelu_erf((h[j][0][1] + h[j][1][1]) + b1.y);
          const float v2 = gelu_erf((h[j][0][2] + h[j][1][2]) + b1.z), v3 = gelu_erf((h[j][0][3] + h[j][1][3]) + b1.w);
          unsigned h0, l0, h1, l1;
          split16_two(v0, v1, h0, l0);
          split16_two(v2, v3, h1, l1);
          // the hidden activation travels as the (unswizzled) split image: row 16 tk + r, chunk 8 hc + w (32 hidden columns = 32 words), high words 8 j + 2 g, low + 16
          const unsigned wo = (kClH + par * 49152u + (unsigned)((16 * tk + r) * 1024 + (8 * hc + wave) * 32 + j * 8 + g * 2)) * 4u;
          xst2(wo, U2{h0, h1});
          xst2(wo + 64u, U2{l0, l1});
        }
        } else {
        // hidden columns 128 hc + 16 w .. + 15: one tile per wave, 8 fragments
        f32x4 ha = zero4, hb = zero4;
        frag(As, kClXs, r, 0, x[0]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) frag(As, kClXs, r, kc + 1, x[(kc + 1) & 1]);
          mma1(kc, x[kc & 1], ha, hb, kc + kClRing < 8);
        }
        const F4 b1 = ld4(sm + kLsL1B + hc * 128 + wave * 16 + g * 4);
        const float v0 = gelu_erf((ha[0] + hb[0]) + b1.x), v1 = gelu_erf((ha[1] + hb[1]) + b1.y);
        const float v2 = gelu_erf((ha[2] + hb[2]) + b1.z), v3 = gelu_erf((ha[3] + hb[3]) + b1.w);
        unsigned h0, l0, h1, l1;
        split16_two(v0, v1, h0, l0);
        split16_two(v2, v3, h1, l1);
        // same image: chunk 4 hc + (w >> 1), tile w & 1 of the chunk
        const unsigned wo = (kClH + par * 49152u + (unsigned)((16 * tk + r) * 1024 + (4 * hc + (wave >> 1)) * 32 + (wave & 1) * 8 + g * 2)) * 4u;
        xst2(wo, U2{h0, h1});
        xst2(wo + 64u, U2{l0, l1});
        }
      }
      CL_STAMP(5);
      publish(kFlagH, epoch);
      refill();
      CL_STAMP(6);
      fresh();
      // ================= E2: the token's hidden activation (64 KB image) from its four members -> Xs region as [16][1032]
      if constexpr (CG == 4) {
        // every wave gathers the slice of ONE producer (wave w <- member (tk, w >> 1), rows 8 (w & 1) .. + 7) as soon as THAT member's flag is up: slices of early
        // producers are in LDS while the last one is still awaited, and there is no barrier between the poll and the loads
        const int prod = wave >> 1;
        wait_one(kFlagH, tk * CG + prod, epoch);
        CL_STAMP(7);
        F4 hv[8];
        int rows[8], uns[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int u = lane + 64 * k8;                 // 16-byte unit of the producer's slice: 8 rows x 64 units
          rows[k8] = 8 * (wave & 1) + (u >> 6);
          uns[k8] = 64 * prod + (u & 63);               // unit of the row's 256 (1024 words)
          hv[k8] = xbuf_ld4(xb, (kClH + par * 49152u + (unsigned)(16 * tk + rows[k8]) * 1024u) * 4u + (unsigned)uns[k8] * 16u);
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
          st4(Xs + rows[k8] * kClHs + (((uns[k8] >> 2) << 4) + (((uns[k8] & 3) ^ ((rows[k8] >> 2) & 3)) << 2)), hv[k8]);
        __syncthreads();
        if (ctl[2] != 0u) { give_up(); return; }
      } else {
      if (!wait_flags(kFlagH, own_mask, epoch)) { give_up(); return; }
      CL_STAMP(7);
      {
        F4 hv[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int q = tid + 512 * k8;                 // 16-byte unit: row q >> 8, unit (q & 255) of the row's 1024 words
          hv[k8] = xbuf_ld4(xb, (kClH + par * 49152u + (unsigned)(16 * tk) * 1024u) * 4u + (unsigned)q * 16u);
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int q = tid + 512 * k8, row = q >> 8, un = q & 255;
          st4(Xs + row * kClHs + (((un >> 2) << 4) + (((un & 3) ^ ((row >> 2) & 3)) << 2)), hv[k8]);
        }
      }
      __syncthreads();
      }
      CL_STAMP(8);
      fresh();
      // ================= Ph3: linear2 for output columns 64 hc + 16 (w & 3) .. + 15, K half w >> 2; halves meet through LDS -> Y
      if constexpr (CG == 4) {
        f32x4 y0 = zero4, y1 = zero4;
        const int kh = wave >> 2;
        F4 x[2][2];
        frag(Xs, kClHs, r, 16 * kh, x[0]);
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
          if (kc + 1 < 16) frag(Xs, kClHs, r, 16 * kh + kc + 1, x[(kc + 1) & 1]);
          mma1(kc, x[kc & 1], y0, y1, kc + kClRing < 16);
        }
        f32x4 y = y0 + y1;
#ifdef CL_TRACE
        asm volatile("" : "+v"(y));
        CL_STAMP(15);
#endif
        if (wave >= 4) *reinterpret_cast<f32x4*>(red2 + ((wave - 4) * 64 + lane) * 4) = y;
        __syncthreads();
        if (wave < 4) {
          y += *reinterpret_cast<const f32x4*>(red2 + (wave * 64 + lane) * 4);
          const F4 b2 = ld4(sm + kLsL2B + hc * 64 + wave * 16 + g * 4), rs = h1_res(hc * 64 + wave * 16 + g * 4);
          xst4((kClY + par * 12288u + (unsigned)((16 * tk + r) * 256 + hc * 64 + wave * 16 + g * 4)) * 4u,
               F4{(y[0] + b2.x) + rs.x, (y[1] + b2.y) + rs.y, (y[2] + b2.z) + rs.z, (y[3] + b2.w) + rs.w});
        }
      } else {
        // output columns 32 hc + 16 (w & 1) .. + 15, K quarter w >> 1 (8 fragments); the quarters meet through LDS in a fixed order
        f32x4 y0 = zero4, y1 = zero4;
        const int tile = wave & 1, kq = wave >> 1;
        F4 x[2][2];
        frag(Xs, kClHs, r, 8 * kq, x[0]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) frag(Xs, kClHs, r, 8 * kq + kc + 1, x[(kc + 1) & 1]);
          mma1(kc, x[kc & 1], y0, y1, kc + kClRing < 8);
        }
        f32x4 y = y0 + y1;
#ifdef CL_TRACE
        asm volatile("" : "+v"(y));
        CL_STAMP(15);
#endif
        if (kq) *reinterpret_cast<f32x4*>(red2 + (((kq - 1) * 2 + tile) * 64 + lane) * 4) = y;
        __syncthreads();
        if (wave < 2) {
#pragma unroll
          for (int q = 0; q < 3; ++q) y += *reinterpret_cast<const f32x4*>(red2 + ((q * 2 + tile) * 64 + lane) * 4);
          const F4 b2 = ld4(sm + kLsL2B + hc * 32 + tile * 16 + g * 4), rs = h1_res(hc * 32 + tile * 16 + g * 4);
          xst4((kClY + par * 12288u + (unsigned)((16 * tk + r) * 256 + hc * 32 + tile * 16 + g * 4)) * 4u,
               F4{(y[0] + b2.x) + rs.x, (y[1] + b2.y) + rs.y, (y[2] + b2.z) + rs.z, (y[3] + b2.w) + rs.w});
        }
      }
      CL_STAMP(9);
      publish(kFlagY, epoch);
      refill();
      CL_STAMP(10);
      fresh();
      // ================= E3 and what follows the layer
      const bool last = l + 1 == p.L, skip_next = !last && l >= nb;
      prm_fetch(last ? 0 : l + 1);
      if (!wait_flags(kFlagY, all_mask, epoch)) { give_up(); return; }
      CL_STAMP(11);      // all twelve even where fewer rows are read (buffer-reuse invariant)
      if (!last && !skip_next) {
        // x' = norm2(y + h1) for all 48 rows -> Xs; input blocks park their own token's rows for the skip connection (cross_attention.py:48-52)
        F4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = y_row(par, wave + 8 * i);
#ifdef CL_TRACE
        asm volatile("" : "+v"(v[0].x), "+v"(v[5].w));
        CL_STAMP(14);
#endif
        ln_rows(v, 6, sm + kLsN2W, sm + kLsN2B);
#pragma unroll
        for (int i = 0; i < 6; ++i) st_row(Xs, kClXs, wave + 8 * i, v[i]);
        if (l < nb) {
          float* pk = p.park + ((size_t)wg * nb + l) * (16 * 256);
#pragma unroll
          for (int i = 0; i < 6; ++i)
            if ((i >> 1) == tk) st4(pk + (unsigned)((wave + 8 * (i & 1)) * 256 + lane * 4), v[i]);
        }
        prm_store(pbuf ^ 1);
        __syncthreads();
      } else if (skip_next) {
        // norm2 of the token's own rows, then x = Linear(cat[x', skip]) for these 16 rows x 64 columns: K half 0 (x') on waves 0-3, half 1 (the parked rows) on waves 4-7
        const int si = l - nb;
        F4 v[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = y_row(par, 16 * tk + wave + 8 * i);
        const float* pk = p.park + ((size_t)wg * nb + (nb - 1 - si)) * (16 * 256);
        const F4 s0v = ld4(pk + (unsigned)(wave * 256 + lane * 4)), s1v = ld4(pk + (unsigned)((wave + 8) * 256 + lane * 4));
        ln_rows(v, 2, sm + kLsN2W, sm + kLsN2B);
        st_row(Xs, kClXs, 16 * tk + wave, v[0]);
        st_row(Xs, kClXs, 16 * tk + wave + 8, v[1]);
        st_row(As, kClXs, wave, s0v);
        st_row(As, kClXs, wave + 8, s1v);
        __syncthreads();
        fresh();
        f32x4 z0 = zero4, z1 = zero4;
        const unsigned zepoch = 16u + (unsigned)(step * nb + si) + 1u, zpar = zepoch & 1u;
        if constexpr (CG == 4) {
        const float* abuf = wave < 4 ? Xs + 16 * tk * kClXs : As;
        F4 x[2][2];
        frag(abuf, kClXs, r, 0, x[0]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (kc + 1 < 8) frag(abuf, kClXs, r, kc + 1, x[(kc + 1) & 1]);
          mma1(kc, x[kc & 1], z0, z1, kc + kClRing < 8);
        }
        f32x4 z = z0 + z1;
        if (wave >= 4) *reinterpret_cast<f32x4*>(red2 + ((wave - 4) * 64 + lane) * 4) = z;
        __syncthreads();
        if (wave < 4) {
          z += *reinterpret_cast<const f32x4*>(red2 + (wave * 64 + lane) * 4);
          const F4 sb = ld4(sm + kLsLayer + hc * 64 + wave * 16 + g * 4);
          xst4((kClZ + zpar * 12288u + (unsigned)((16 * tk + r) * 256 + hc * 64 + wave * 16 + g * 4)) * 4u, F4{z[0] + sb.x, z[1] + sb.y, z[2] + sb.z, z[3] + sb.w});
        }
        } else {
        // 32 columns per member: tile w & 1, K quarter w >> 1 of the 512 (quarters 0, 1: x', 2, 3: the parked rows), 4 fragments
        const int tile = wave & 1, kq = wave >> 1;
        const float* abuf = kq < 2 ? Xs + 16 * tk * kClXs : As;
        F4 x[2][2];
        frag(abuf, kClXs, r, 4 * (kq & 1), x[0]);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          if (kc + 1 < 4) frag(abuf, kClXs, r, 4 * (kq & 1) + kc + 1, x[(kc + 1) & 1]);
          mma1(kc, x[kc & 1], z0, z1, kc + kClRing < 4);
        }
        f32x4 z = z0 + z1;
        if (kq) *reinterpret_cast<f32x4*>(red2 + (((kq - 1) * 2 + tile) * 64 + lane) * 4) = z;
        __syncthreads();
        if (wave < 2) {
#pragma unroll
          for (int q = 0; q < 3; ++q) z += *reinterpret_cast<const f32x4*>(red2 + ((q * 2 + tile) * 64 + lane) * 4);
          const F4 sb = ld4(sm + kLsLayer + hc * 32 + tile * 16 + g * 4);
          xst4((kClZ + zpar * 12288u + (unsigned)((16 * tk + r) * 256 + hc * 32 + tile * 16 + g * 4)) * 4u, F4{z[0] + sb.x, z[1] + sb.y, z[2] + sb.z, z[3] + sb.w});
        }
        }
        publish(kFlagZ, zepoch);
      refill();
        fresh();
        if (!wait_flags(kFlagZ, all_mask, zepoch)) { give_up(); return; }
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = xbuf_ld4(xb, (kClZ + zpar * 12288u + (unsigned)((wave + 8 * i) * 256 + lane * 4)) * 4u);
#pragma unroll
        for (int i = 0; i < 6; ++i) st_row(Xs, kClXs, wave + 8 * i, v[i]);
        prm_store(pbuf ^ 1);
        __syncthreads();
      } else {
        // end of the step, every member for itself: norm2 + encoder.norm of the latent token's rows w (unconditional) and w + 8 (conditional) of motion w,
        // CFG (mld.py:339-342), DDIM eta = 0 (mld.py:345-346), the next step's rows
        F4 v[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = y_row(par, wave + 8 * i);
        ln_rows(v, 2, sm + kLsN2W, sm + kLsN2B);
        ln_rows(v, 2, sm_fin, sm_fin + 256);
        const float sat = p.ddim[step * 4], s1mat = p.ddim[step * 4 + 1], sap = p.ddim[step * 4 + 2], s1map = p.ddim[step * 4 + 3];
        float* lp = lats + wave * 256 + lane * 4;
        const F4 xt = ld4(lp);
        const float eu[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, ec[4] = {v[1].x, v[1].y, v[1].z, v[1].w}, xtv[4] = {xt.x, xt.y, xt.z, xt.w};
        float nv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float eps = eu[i] + p.guidance * (ec[i] - eu[i]);
          const float x0 = (xtv[i] - s1mat * eps) / sat;
          nv[i] = sap * x0 + s1map * eps;
        }
        st4(lp, F4{nv[0], nv[1], nv[2], nv[3]});
        prm_store(pbuf ^ 1);
        if (step + 1 < p.n) assemble(step + 1);      // (reads this wave's own latent row only)
        __syncthreads();
      }
      CL_STAMP(12);
      pbuf ^= 1;
    }
  }
#ifdef CL_TRACE
  if (p.trace && lane == 0) {
    unsigned long long* o = p.trace + ((size_t)blockIdx.x * 8 + wave) * 16;
    for (int k = 0; k < 16; ++k) o[k] = ph[k];
  }
#endif
  if (member == 0) {
    const int c = tid >> 6, c4 = tid & 63;
    if (s0 + c < p.s_end) st4(p.lat + (long long)(s0 + c) * 256 + c4 * 4, ld4(lats + c * 256 + c4 * 4));
  }
  // Every polled word goes back to zero before the launch ends: a member that is past its last wait counts itself in word 28 of the Z line; the
  // last arrival polls nothing any more and neither does anybody else, so it clears the cluster's flag words.  clear_cluster_flags_kernel in front of
  // the launch (Guideline 16 "Re-initialise every call") does it again.
  __syncthreads();
  if (tid == 0) {
#if defined(MLDHIP_SIM)
    const unsigned prev = flags[kFlagZ * kClFlagLine + 28]++;
#else
    const unsigned prev = __hip_atomic_fetch_add(flags + kFlagZ * kClFlagLine + 28, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (prev == (unsigned)kM - 1u)
      for (int i = 0; i < kClFlagWords; ++i) flag_store(flags + i, 0u);
  }
}

}  // namespace mld
