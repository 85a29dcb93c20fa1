// Latency-oriented GEMM for the denoiser loop (M = 6B token rows, a few hundred at most).
//
// Measured on MI355X (profiles/r01_v1_*): with M = 384 every GEMM of the 50-step loop is bound by
// the memory round trip (~1-2 us to L2/MALL), not by MFMA or bandwidth, so the design goal is
// "ONE round trip per kernel": a workgroup issues every global load it will ever need up front
// (32 x 256 A rows + 64 x 256 weight rows, one coalesced 1 KiB row per wave instruction), parks the
// tile in LDS, and only then runs its 64 MFMAs per wave.  K is handled in slices of 256 (blockIdx.z);
// a GEMM with K > 256 leaves raw fp32 partial slabs and the CONSUMER kernel sums them while it
// assembles its own A tile.  That same "A prologue" also applies bias, residual and the post-norm
// LayerNorm (one wave owns one 256-wide row: statistics are two wave reductions, no LDS, no extra
// launch) and, for the out-projection, computes the 3-token self-attention on the fly.  Net effect per
// encoder layer (cross_attention.py:259-272): 4 launches (QKV, out-proj, FFN1, FFN2) instead of the
// ~30 ATen kernels of the reference or the 5 + separate-LN launches of the first version here.
//
// Row layout: token-major, row = s*R + r (s = token 0..2, r = sample), 256 floats per row.
#pragma once
#include "elementwise.hpp"
#include "rt.hpp"

namespace mld {

struct ASrc {
  const float* base = nullptr;   // plain rows, slab base (nsplit > 0) or packed qkv (attn_R > 0)
  int ld = 0;                    // row stride in floats
  int nsplit = 0;                // > 0: row = sum of nsplit slabs (+ bias + res), each [M][256]
  long long pstride = 0;         // slab stride in floats
  const float* bias = nullptr;   // [256], combine mode
  const float* res = nullptr;    // residual rows [M][ldres], combine mode (optional)
  int ldres = 0;
  const float* gamma = nullptr;  // LayerNorm(gamma, beta) after the sum when non-null
  const float* beta = nullptr;
  float* out = nullptr;          // write the assembled rows back ([M][ldout]; done by blockIdx.y == 0 only)
  int ldout = 0;
  int attn_R = 0;                // > 0: rows are 3-token attention outputs computed from qkv[3R][768]
};

struct Tile32Args {
  ASrc src[2];
  int nz0 = 1;                   // K slices [0, nz0) read src[0] at column 256*z, the rest src[1] at 256*(z-nz0)
  const float* W = nullptr;      // [N][ldw] (nn.Linear layout), slice z uses columns 256*z ..
  int ldw = 0;
  const float* bias = nullptr;   // direct epilogue: Y = act(acc + bias)
  int act = 0;                   // 0 none, 1 erf-GELU, 2 SiLU
  float* Y = nullptr;
  int ldy = 0;
  float* P = nullptr;            // partial epilogue when non-null: P[z][M][N] = acc (raw)
  long long pstride = 0;
  int M = 0, N = 0;
  int w_split = 0;               // PREC_BF16X3: W points into the pre-split (hi | lo half) image of the weight arena (elementwise.hpp)
  unsigned long long* trace = nullptr;   // measurement only: 8 timestamps per wave (see mldhip_profile_trace)
};

constexpr int kT32Stride = 264;                         // LDS row stride (floats): 256 + 8 pad (= 8 mod 16: conflict-free ds_read_b128, gemm.hpp kGemmLdsStride)
constexpr int kT32LdsFloats = (32 + 64) * kT32Stride + 32;   // A tile + W tile (+ 32 spare words: the per-row scales of the fp8 mode retired in round 6)
constexpr int kT32LdsBytes = kT32LdsFloats * 4;              // 101,504 B -> one workgroup per CU

// ---- operand formats shared by the loop kernels (tile32 / strip): an LDS row holds 256 K-values of one A or W row as
// fp32 (256 words) or 16-bit formats (128 / 2 x 128 words); lane l of the storing wave owns k = 4l..4l+3; a fragment of the
// 32-wide K chunk kc is lane (r, g)'s 8 values -- k = 32kc + 8g .. + 7 in the packed formats (rt.hpp MFMA operand layouts),
// k = 32kc + 4g .. + 3 and 32kc + 16 + 4g .. + 3 in fp32 (any pairing is legal when A and W agree; this one is LDS-conflict free).
template <int PREC>
__device__ __forceinline__ void st_operand(float* row, int lane, F4 v) {
  if constexpr (PREC == PREC_F32) {
    st4(row + lane * 4, v);
  } else if constexpr (PREC == PREC_BF16) {
    *reinterpret_cast<U2*>(reinterpret_cast<unsigned*>(row) + lane * 2) = U2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  } else {
    static_assert(PREC == PREC_BF16X3, "operand format");
    // split-f16 image, the one split_bf16_weights_kernel writes: per 32-wide K chunk 16 words of high halves, then 16 of low halves
    unsigned h0, l0, h1, l1;
    split16_pair(v.x, v.y, h0, l0);
    split16_pair(v.z, v.w, h1, l1);
    unsigned* d = reinterpret_cast<unsigned*>(row) + (lane >> 3) * 32 + (lane & 7) * 2;
    *reinterpret_cast<U2*>(d) = U2{h0, h1};
    *reinterpret_cast<U2*>(d + 16) = U2{l0, l1};
  }
}
// one 32-wide K chunk of a 16x16 tile: acc += A_frag . B_frag (fp32: 8 MFMAs alternating over two accumulators to hide
// the dependent-issue latency; bf16: one MFMA, accumulators alternate by chunk parity; split-f16: three MFMAs, the two
// cross terms in acc0 and the hi x hi term in acc1)
template <int PREC>
__device__ __forceinline__ void mma_chunk(const float* arow, const float* wrow, int kc, int g, f32x4& acc0, f32x4& acc1) {
  if constexpr (PREC == PREC_BF16X3) {
    const U4* ar = reinterpret_cast<const U4*>(arow) + kc * 8 + g;
    const U4* wr = reinterpret_cast<const U4*>(wrow) + kc * 8 + g;
    const U4 ah = ar[0], al = ar[4], wh = wr[0], wl = wr[4];
    acc0 = mfma_x3_16x16x32(al, wh, acc0);
    acc1 = mfma_x3_16x16x32(ah, wh, acc1);
    acc0 = mfma_x3_16x16x32(ah, wl, acc0);
  } else if constexpr (PREC == PREC_F32) {
    const F4 a0 = ld4(arow + kc * 32 + g * 4), a1 = ld4(arow + kc * 32 + 16 + g * 4);   // k-slots 4g .. + 3 and 16 + 4g .. + 3 (A and W alike)
    const F4 b0 = ld4(wrow + kc * 32 + g * 4), b1 = ld4(wrow + kc * 32 + 16 + g * 4);
    acc0 = mfma_f32_16x16x4(a0.x, b0.x, acc0);
    acc1 = mfma_f32_16x16x4(a0.y, b0.y, acc1);
    acc0 = mfma_f32_16x16x4(a0.z, b0.z, acc0);
    acc1 = mfma_f32_16x16x4(a0.w, b0.w, acc1);
    acc0 = mfma_f32_16x16x4(a1.x, b1.x, acc0);
    acc1 = mfma_f32_16x16x4(a1.y, b1.y, acc1);
    acc0 = mfma_f32_16x16x4(a1.z, b1.z, acc0);
    acc1 = mfma_f32_16x16x4(a1.w, b1.w, acc1);
  } else {
    static_assert(PREC == PREC_BF16, "operand format");
    const U4 a = reinterpret_cast<const U4*>(arow)[kc * 4 + g], b = reinterpret_cast<const U4*>(wrow)[kc * 4 + g];
    if (kc & 1) acc1 = mfma_bf16_16x16x32(a, b, acc1);
    else acc0 = mfma_bf16_16x16x32(a, b, acc0);
  }
}

// Independent wave reductions issued back to back: the DPP chains of different rows interleave.
template <int N>
__device__ __forceinline__ void sum64xn(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = sum64(v[i]);
}
template <int N>
__device__ __forceinline__ void sum16xn(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = sum16(v[i]);
}
__device__ __forceinline__ F4 f4add(F4 a, F4 b) { return F4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// grid = (ceil(M/MT), N/64, K/256); block = 512 (8 waves).
//   MT = 32: wave w -> column tile w&3, row tile w>>2, all of K (64 MFMAs per wave).
//   MT = 16: wave w -> column tile w&3, K half w>>2 (32 MFMAs per wave), halves summed through LDS.  Used for
//            the N = 256 GEMMs (out-proj, skip linear) which would otherwise run on 48 / 96 workgroups: twice
//            the workgroups, half the A bytes and half the per-SIMD MFMA queue per workgroup.
// A prologue: each wave assembles MT/8 rows (w, w+8, ... of the tile); lane l owns columns 4l..4l+3.
// EVERY global load of the workgroup is issued before the first dependent instruction.
// NS0 = compile-time slab count of src[0] when it is a combine source (0: src[0] is plain / attention):
// keeps every load unconditional and straight-line (a per-load `cond ? load : 0` makes hipcc branch and
// wait per element -- cdna_hip_programming.md, "three .s-level traps" (c)).
// (Passing K through LDS in two 128-wide pieces so that two workgroups fit a CU was measured: +2.9 % with four batches in
// flight, -10 % for one batch; removed -- kernels/strip.hpp is the throughput form.  profiles/r01_v17_xcd_kh_ab.txt.)
// PREC: operand format of the MFMAs (rt.hpp PREC_F32 / PREC_BF16 / PREC_BF16X3 = split-f16, 3 MFMAs of 16 cycles per 32-wide
// K chunk instead of 8 of 32; the A prologue and the epilogue stay fp32).
// MODE: how the A rows are obtained is a COMPILE-TIME property of the launch.  NS0 = 0: MODE 0 = plain rows (src[0] / src[1] by K slice),
// MODE 1 = 3-token attention outputs; NS0 > 0: MODE 0 = every K slice combines src[0]'s slabs, MODE 1 = slices >= nz0 read src[1] as
// plain rows (the skip linear).  As run-time branches the three source paths cost a second memory round trip per kernel: hipcc lays
// them out as successors of each other (an `s_cbranch_execz` skip edge), so the wait-count pass sees the OTHER path's row loads as
// pending writes of the registers this path re-uses and waits `vmcnt(0)` -- i.e. for the weight loads issued at the top -- before it
// issues its own loads (r03: FFN1 prologue 5 230 cycles against 2 152 for the plain-row FFN2; profiles/r03_tile32_phase_trace.json).
template <int MT, int NS0, bool TRACE, int PREC = PREC_F32, int MODE = 0>
__global__ __launch_bounds__(512, 2) void gemm_tile32_kernel(Tile32Args p) {
  static_assert(MT == 16 || MT == 32, "row tile");
  static_assert(MODE == 0 || MODE == 1, "source mode");
  constexpr bool kAttn = NS0 == 0 && MODE == 1;       // A rows = attention outputs
  constexpr bool kTwo = NS0 > 0 && MODE == 1;         // combine source + a plain second source
  static_assert(PREC == PREC_F32 || PREC == PREC_BF16 || PREC == PREC_BF16X3, "operand format");
  constexpr int RPW = MT / 8;                 // A rows assembled per wave
  constexpr int KW = 256, ST = kT32Stride;    // K columns resident in LDS, LDS row stride (floats)
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* As = smem;                          // [MT][ST]
  float* Ws = smem + MT * ST;                // [64][ST]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (An XCD-aware tile order -- block b runs on XCD b % 8; give each XCD a contiguous range of column tiles so that it
  // pulls 1/8 of the weight panel instead of all of it -- was measured SLOWER: 6 825 vs 7 260 motions/s at 4 batches in
  // flight, 3 397 vs 3 728 single-stream.  All workgroups of an XCD then hammer the same few L2 channels at the same
  // instant; in dispatch order the reads spread over every channel of every XCD.  profiles/r01_v17_xcd_kh_ab.txt.)
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int m0 = bx * MT, n0 = by * 64, z = bz;
  const bool second = (NS0 == 0 || kTwo) ? z >= p.nz0 : false;
  const ASrc& src = second ? p.src[1] : p.src[0];
  const int acol = (second ? z - p.nz0 : z) * 256;
  const int wcol = z * 256;
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, rt0 = 0;
  constexpr bool tracing = TRACE;
  if constexpr (tracing) { rt0 = realtime_100mhz(); ts[0] = clock_pinned(); }

  // ---- weights: 8 rows per wave, one coalesced 1 KiB row per instruction
  F4 wreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int n = n0 + wave + i * 8;
    n = n < p.N ? n : p.N - 1;
    wreg[i] = ld4(p.W + (long long)n * p.ldw + wcol + lane * 4);
  }
  // epilogue bias of this wave's output column, fetched now so its latency hides behind everything else
  const int ecol = n0 + (wave & 3) * 16 + (lane & 15);
  float ebias = 0.f;
  if (p.bias) ebias = p.bias[ecol < p.N ? ecol : p.N - 1];
  int rows[RPW];
  bool live[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = m0 + wave + i * 8;
    live[i] = row < p.M;
    rows[i] = live[i] ? row : p.M - 1;
  }
  F4 areg[RPW];
  if constexpr (kAttn) {
    // nn.MultiheadAttention over the 3 tokens of one sample (cross_attention.py:265-266): head = lane >> 4
    // (64 dims = 16 lanes x 4), q pre-scaled by 1/sqrt(64), softmax over the 3 keys, all in registers.
    const int R = src.attn_R;
    F4 q[RPW], k[RPW][3], v[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int tok = rows[i] / R, smp = rows[i] - tok * R;
      q[i] = ld4(src.base + (long long)rows[i] * 768 + lane * 4);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float* kr = src.base + (long long)(j * R + smp) * 768 + 256 + lane * 4;
        k[i][j] = ld4(kr);
        v[i][j] = ld4(kr + 256);
      }
    }
    float sc[RPW * 3];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float d = q[i].x * k[i][j].x;
        d = fmaf(q[i].y, k[i][j].y, d);
        d = fmaf(q[i].z, k[i][j].z, d);
        d = fmaf(q[i].w, k[i][j].w, d);
        sc[i * 3 + j] = d;
      }
    sum16xn<RPW * 3>(sc);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const float s0 = sc[i * 3] * 0.125f, s1 = sc[i * 3 + 1] * 0.125f, s2 = sc[i * 3 + 2] * 0.125f;
      const float m = fmaxf(s0, fmaxf(s1, s2));
      const float e0 = expf(s0 - m), e1 = expf(s1 - m), e2 = expf(s2 - m);
      const float inv = 1.0f / (e0 + e1 + e2);
      const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
      areg[i].x = p0 * v[i][0].x + p1 * v[i][1].x + p2 * v[i][2].x;
      areg[i].y = p0 * v[i][0].y + p1 * v[i][1].y + p2 * v[i][2].y;
      areg[i].z = p0 * v[i][0].z + p1 * v[i][1].z + p2 * v[i][2].z;
      areg[i].w = p0 * v[i][0].w + p1 * v[i][1].w + p2 * v[i][2].w;
    }
  } else if (NS0 == 0 || (kTwo && second)) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) areg[i] = ld4(src.base + (long long)rows[i] * src.ld + acol + lane * 4);
  } else {
    // ---- combine: sum of NS0 slabs + bias (+ residual), then optional LayerNorm; one wave owns a row.
    constexpr int NS = NS0 > 0 ? NS0 : 1;
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
    for (int i = 0; i < RPW; ++i) {   // ((s0+s1)+s2)+s3, + bias, + res
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
    if (src.out && by == 0) {
#pragma unroll
      for (int i = 0; i < RPW; ++i)
        if (live[i]) st4(src.out + (long long)rows[i] * src.ldout + lane * 4, areg[i]);
    }
  }
  if constexpr (tracing) ts[1] = clock_pinned();      // every load landed, prologue math done
  // ---- one 16x16 tile per wave; two accumulators hide the MFMA dependent latency
  const int r = lane & 15, g = lane >> 4;
  const int ct = wave & 3;
  const int rt = MT == 32 ? (wave >> 2) : 0;          // row tile (MT = 32)
  const int kh = MT == 16 ? (wave >> 2) : 0;          // K half of the resident piece (MT = 16)
  constexpr int KCH = MT == 32 ? 8 : 4;               // 32-wide K chunks per wave
  const float* ap = As + (rt * 16 + r) * ST;
  const float* wp = Ws + (ct * 16 + r) * ST;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (PREC == PREC_BF16X3) break;
      st_operand<PREC>(Ws + (wave + i * 8) * ST, lane, wreg[i]);
    }
    if constexpr (PREC == PREC_BF16X3) {
      if (p.w_split) {                          // W came from the pre-split image: the loaded words ARE the row image
#pragma unroll
        for (int i = 0; i < 8; ++i) st4(Ws + (wave + i * 8) * ST + lane * 4, wreg[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) st_operand<PREC>(Ws + (wave + i * 8) * ST, lane, wreg[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) st_operand<PREC>(As + (wave + i * 8) * ST, lane, areg[i]);
    if constexpr (tracing) ts[2] = clock_pinned();      // tile parked in LDS (this wave)
    __syncthreads();
    if constexpr (tracing) ts[3] = clock_pinned();      // barrier passed
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) mma_chunk<PREC>(ap, wp, kh * KCH + kc, g, acc0, acc1);
  }
  f32x4 acc = acc0 + acc1;
  if constexpr (MT == 16) {
    // combine the two K halves: upper waves park their tile in LDS (the W panel is dead after a barrier)
    __syncthreads();
    float* red = Ws;                                   // [4 col tiles][64 lanes][4]
    if (kh == 1) *reinterpret_cast<f32x4*>(red + (ct * 64 + lane) * 4) = acc;
    __syncthreads();
    if (kh == 0) acc += *reinterpret_cast<const f32x4*>(red + (ct * 64 + lane) * 4);
  }
  if constexpr (tracing) { asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); ts[4] = clock_pinned(); }   // MFMAs retired
  // ---- epilogue: direct stores (16 lanes write 64 contiguous bytes per row).  A coalesced variant that
  //      transposed the tile through LDS was measured SLOWER (+0.7-1.8 k cycles: two barriers + an LDS
  //      round trip cost more than the wider stores save; profiles/r01_v5).
  const int col = n0 + ct * 16 + r;
  if (col < p.N && kh == 0) {
    if (p.P) {
      float* P = p.P + z * p.pstride;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + rt * 16 + g * 4 + i;
        if (row < p.M) P[(long long)row * p.N + col] = acc[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + rt * 16 + g * 4 + i;
        if (row < p.M) {
          float v = acc[i] + ebias;
          if (p.act == 1) v = gelu_erf(v);
          else if (p.act == 2) v = silu(v);
          p.Y[(long long)row * p.ldy + col] = v;
        }
      }
    }
  }
  if constexpr (tracing) {
    ts[5] = clock_pinned();                 // epilogue stores issued and drained
    if (lane == 0) {
      const long long wg = bx + (long long)gridDim.x * (by + (long long)gridDim.y * bz);
      unsigned long long* o = p.trace + (wg * 8 + wave) * 8;
      for (int i = 0; i < 6; ++i) o[i] = ts[i];
      o[6] = rt0;
      o[7] = realtime_100mhz();
    }
  }
}

// ----------------------------------------------------------------------------------------------
// End of one reverse step.  Token-0 rows only: x = LN2(sum FFN2 slabs + b2 + h1) (last layer's norm2),
// e = LN_final(x) (cross_attention.py:62-63, mld_denoiser.py:206), CFG (mld.py:339-342), DDIM step
// (diffusers DDIMScheduler.step, eta 0), and the next step's token-0 / time rows.  grid = B, block = 256.
struct FinalArgs {
  const float* P; int nsplit; long long pstride;   // last FFN2 partial slabs [nsplit][3R][256]
  const float* b2; const float* H1;                // FFN2 bias, residual rows (norm1 output)
  const float* g2; const float* be2;               // last layer norm2
  const float* gf; const float* bef;               // encoder.norm
};

__device__ __forceinline__ float final_row_value(const FinalArgs& f, long long row, int d, float* sh) {
  float x = f.b2[d] + f.H1[row * 256 + d];
  for (int z = 0; z < f.nsplit; ++z) x += f.P[z * f.pstride + row * 256 + d];
  float mean = block_sum_256(x, sh, d) * (1.0f / 256.0f);
  float xc = x - mean;
  float var = block_sum_256(xc * xc, sh, d) * (1.0f / 256.0f);
  x = xc * rsqrtf(var + kLnEps) * f.g2[d] + f.be2[d];
  mean = block_sum_256(x, sh, d) * (1.0f / 256.0f);
  xc = x - mean;
  var = block_sum_256(xc * xc, sh, d) * (1.0f / 256.0f);
  return xc * rsqrtf(var + kLnEps) * f.gf[d] + f.bef[d];
}

__global__ __launch_bounds__(256) void den_final_step_kernel(FinalArgs f, float* __restrict__ lat, float* __restrict__ X0,
                                                             const float* __restrict__ pe0, const float* __restrict__ t1_next,
                                                             int B, float guidance, DdimCoef c) {
  __shared__ float sh[4];
  const int b = blockIdx.x, d = threadIdx.x, R = 2 * B;
  const float eu = final_row_value(f, b, d, sh);
  const float ec = final_row_value(f, B + b, d, sh);
  const float eps = eu + guidance * (ec - eu);
  const float x = lat[(long long)b * 256 + d];
  const float x0 = (x - c.sqrt_1mat * eps) / c.sqrt_at;
  const float xn = c.sqrt_ap * x0 + c.sqrt_1map * eps;
  lat[(long long)b * 256 + d] = xn;
  const float tok = xn + pe0[d];
  X0[(long long)b * 256 + d] = tok;
  X0[(long long)(B + b) * 256 + d] = tok;
  if (t1_next) {
    const float tt = t1_next[d];
    X0[(long long)(R + b) * 256 + d] = tt;
    X0[(long long)(R + B + b) * 256 + d] = tt;
  }
}

// Stand-alone MldDenoiser.forward output: out[r] = LN_final(LN2(...)) for the R token-0 rows.  grid = R.
__global__ __launch_bounds__(256) void den_final_rows_kernel(FinalArgs f, float* __restrict__ out) {
  __shared__ float sh[4];
  const int r = blockIdx.x, d = threadIdx.x;
  out[(long long)r * 256 + d] = final_row_value(f, r, d, sh);
}

}  // namespace mld
