// Throughput-oriented GEMM for the denoiser loop when several bs-64 requests run as ONE chain (M = 6B >= ~768 token
// rows): Y[32 x 64] per workgroup, the A strip (32 rows x all of K) resident in LDS after the same "A prologue" the
// latency kernels use (split-K slab sum + bias + residual + post-norm LayerNorm, or the 3-token self-attention,
// kernels/tile32.hpp), the weight panel streamed through a double-buffered LDS ring in 32-wide K chunks behind a
// 4-deep register prefetch ring, so chunk k's MFMAs run while chunks k+1..k+4 are in flight.
//
// Why a second kernel next to tile32.hpp: tile32 parks A AND W whole in LDS (99.8 KB -> one 8-wave workgroup per CU,
// every load issued before the first MFMA): right when one launch has <= 192 workgroups and the only goal is a short
// dependent chain (M = 384), but at M = 1 536 its 576-768 workgroups run in three serial rounds with no overlap of one
// round's loads and another's MFMAs (measured r01: 24 TF on FFN1).  Here a workgroup holds 52-70 KB of LDS and <= 128
// VGPRs per lane, so two or three are resident per CU; with 8 waves each (one 16-row tile per wave: +4.5 % end to end over
// 4 waves with two tiles each, profiles/r02_strip_options_ab.json) every SIMD has four waves to hide LDS and global latency.
// (A 64 x 64 / 8-wave tile for FFN2 instead of 32 x 64 / 4 waves: 14.4 -> 13.5 us alone, nothing end to end; not kept.)
//
// Row layout and the meaning of every ASrc / Tile32Args field are those of tile32.hpp (token-major rows, 256 floats).
// Replaces the same reference code: cross_attention.py:259-272 (encoder layer), :56-58 (skip linear).
#pragma once
#include "tile32.hpp"

namespace mld {

// LDS row strides = 8 (mod 16) words and fp32 fragments read as words 4g .. 4g + 3 and 16 + 4g .. 16 + 4g + 3 of the chunk (k-slot
// pairing is free as long as A and W agree): a ds_read_b128 is served in four 16-lane groups over 64 banks, and "row r, word 4g" is
// conflict free at these strides while the former "row r, words 8g / 8g + 4" is 2-way at every stride (gemm.hpp kGemmLdsStride).
constexpr int kStripWStride = 40;                        // floats per staged W row: one 32-wide K chunk + 8 pad
template <int NSRC, int CT = 1>
constexpr int strip_lds_bytes() { return (32 * (256 * NSRC + 8) + 2 * 64 * CT * kStripWStride + 32) * 4; }   // CT=1: 54 400 / 87 168 B; CT=2: 74 880 B (two workgroups per CU)

// one 32-wide K chunk: RT 16-row tiles of the A strip against CT 16-column weight tiles (operand formats: tile32.hpp).
// All fragments are read first, then the MFMAs run interleaved over the RT x CT independent accumulators, so consecutive
// matrix instructions never depend on each other.  `a` = row r of the first row tile (next tile 16 * ast words on),
// `w` = row r of the first column tile of the current LDS buffer (next tile 64 * wst words on).
template <int PREC, int RT, int CT>
__device__ __forceinline__ void strip_mma(const float* a, int ast, const float* w, int wst, int kc, int g, f32x4 (&acc)[CT][RT]) {
  if constexpr (PREC == PREC_F32) {
    F4 x[RT][2], y[CT][2];
#pragma unroll
    for (int t = 0; t < RT; ++t) { x[t][0] = ld4(a + t * 16 * ast + kc * 32 + g * 4); x[t][1] = ld4(a + t * 16 * ast + kc * 32 + 16 + g * 4); }
#pragma unroll
    for (int c = 0; c < CT; ++c) { y[c][0] = ld4(w + c * 64 * wst + g * 4); y[c][1] = ld4(w + c * 64 * wst + 16 + g * 4); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[c][t] = mfma_f32_16x16x4(x[t][h].x, y[c][h].x, acc[c][t]);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[c][t] = mfma_f32_16x16x4(x[t][h].y, y[c][h].y, acc[c][t]);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[c][t] = mfma_f32_16x16x4(x[t][h].z, y[c][h].z, acc[c][t]);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[c][t] = mfma_f32_16x16x4(x[t][h].w, y[c][h].w, acc[c][t]);
    }
  } else {
    static_assert(PREC == PREC_BF16, "operand format");
    U4 x[RT], y[CT];
#pragma unroll
    for (int t = 0; t < RT; ++t) x[t] = reinterpret_cast<const U4*>(a + t * 16 * ast)[kc * 4 + g];
#pragma unroll
    for (int c = 0; c < CT; ++c) y[c] = reinterpret_cast<const U4*>(w + c * 64 * wst)[g];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[c][t] = mfma_bf16_16x16x32(x[t], y[c], acc[c][t]);
  }
}

// grid = (ceil(M/32), ceil(N/(64 CT))); block = 64 NW.  NW = 4: wave w owns output columns [16w + 64c, 16w + 64c + 16), c < CT,
// of the tile, both 16-row tiles.  NW = 8: the same columns for waves w and w + 4, which take one 16-row tile each -- half the
// dependent MFMA chain and half the prologue rows per wave, four waves per SIMD with two workgroups per CU.  CT = 2 (a 32 x 128 tile, 70 KB of LDS, two workgroups per CU) halves the number of
// workgroups that repeat the same A prologue and keeps N = 1024 at 1 920 rows within one resident round.  K = 256 * NSRC: columns [0, 256) come from src[0] (plain rows, combine, or attention), columns
// [256, 512) from src[1] (plain rows; the skip connection's second K segment).
// NS0 = compile-time slab count of src[0] in combine mode (0: plain rows or attention).
// PREC = operand format of the MFMAs (rt.hpp); prologue, accumulation and epilogue are fp32 in every mode.
// ACT  = epilogue activation of the direct (Y) output, compile time: 0 none, 1 erf-GELU (FFN1).  Besides saving a branch it
//        gives QKV and FFN1 -- same prologue, same K -- distinct kernel names in a profile.
// TRACE = measurement build (mldhip_profile_trace): six shader-clock stamps per wave, see the end of the kernel.
template <int NS0, int NSRC, bool ATTN, int PREC = PREC_F32, int ACT = 0, int CT = 1, int NW = 4, bool TRACE = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? ((ATTN || NSRC == 2) ? 2 : 4) : (ATTN || CT == 2 || NS0 >= 2) ? 2 : 3) void gemm_strip_kernel(Tile32Args p) {
  static_assert(CT == 1 || CT == 2, "one or two 16-column tiles per wave");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int BN = 64 * CT, RT = NW == 8 ? 1 : 2, WJ = BN / (NW * 8);
  static_assert(NSRC == 1 || NSRC == 2, "one or two 256-wide K segments");
  static_assert(!(ATTN && (NS0 != 0 || NSRC != 1)), "the attention prologue feeds the out-projection only");
  constexpr int K = 256 * NSRC, ST = K + 8, KCS = K / 32, RPW = 32 / NW;
  constexpr int RD = 4;   // register prefetch ring depth in chunks (8 = the whole K = 256 panel up front was measured 1 % slower end to end)
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* As = smem;                          // [32][ST]
  float* Ws = smem + 32 * ST;                // [2][BN][kStripWStride]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, rt0 = 0;
  if constexpr (TRACE) { rt0 = realtime_100mhz(); ts[0] = clock_pinned(); }
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * BN;
  const ASrc& src = p.src[0];

  // ---- weight ring: thread t stages rows (t>>3) + 8 NW j, j < WJ, of the panel, 16 bytes at column 4*(t&7) of the chunk
  const int wrow = tid >> 3, wc4 = tid & 7;
  const float* wptr[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    int n = n0 + wrow + 8 * NW * j;
    n = n < p.N ? n : p.N - 1;
    wptr[j] = p.W + (long long)n * p.ldw + wc4 * 4;
  }
  F4 ring[RD][WJ];
  auto gload = [&](int c) {                  // c is a constant after unrolling
#pragma unroll
    for (int j = 0; j < WJ; ++j) ring[c % RD][j] = ld4(wptr[j] + c * 32);
  };
  auto gload_first = [&]() {
#pragma unroll
    for (int c = 0; c < RD; ++c) gload(c);
  };
  auto lstore = [&](int c) {
    float* dst = Ws + (c & 1) * BN * kStripWStride;
#pragma unroll
    for (int j = 0; j < WJ; ++j)           // "lane" = the 16-byte slot within the chunk
      st_operand<PREC>(dst + (wrow + 8 * NW * j) * kStripWStride, wc4, ring[c % RD][j]);
  };

  // epilogue bias of this lane's output columns, fetched now (clamped, unconditional)
  float ebias[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int ecol = n0 + ((wave & 3) + 4 * c) * 16 + (lane & 15);
    ebias[c] = 0.f;
    if (p.bias) ebias[c] = p.bias[ecol < p.N ? ecol : p.N - 1];
  }

  // ---- A prologue: wave w assembles rows w, w + NW, ... of the strip; lane l owns columns 4l..4l+3
  int rows[RPW];
  bool live[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = m0 + wave + i * NW;
    live[i] = row < p.M;
    rows[i] = live[i] ? row : p.M - 1;
  }
  F4 areg[RPW];
  if constexpr (ATTN) {
    // 3-token self-attention on load (tile32.hpp, cross_attention.py:265-266), four rows at a time to bound registers;
    // the weight ring is started once the last pass's loads are out
    const int R = src.attn_R;
    constexpr int NP = RPW / 4;              // passes of four rows
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      F4 q[4], k[4][3], v[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rows[h * 4 + i], tok = row / R, smp = row - tok * R;
        q[i] = ld4(src.base + (long long)row * 768 + lane * 4);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float* kr = src.base + (long long)(j * R + smp) * 768 + 256 + lane * 4;
          k[i][j] = ld4(kr);
          v[i][j] = ld4(kr + 256);
        }
      }
      if (h == NP - 1) gload_first();
      float sc[12];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float d = q[i].x * k[i][j].x;
          d = fmaf(q[i].y, k[i][j].y, d);
          d = fmaf(q[i].z, k[i][j].z, d);
          d = fmaf(q[i].w, k[i][j].w, d);
          sc[i * 3 + j] = d;
        }
      sum16xn<12>(sc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float s0 = sc[i * 3] * 0.125f, s1 = sc[i * 3 + 1] * 0.125f, s2 = sc[i * 3 + 2] * 0.125f;
        const float m = fmaxf(s0, fmaxf(s1, s2));
        const float e0 = expf(s0 - m), e1 = expf(s1 - m), e2 = expf(s2 - m);
        const float inv = 1.0f / (e0 + e1 + e2);
        const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
        F4 o;
        o.x = p0 * v[i][0].x + p1 * v[i][1].x + p2 * v[i][2].x;
        o.y = p0 * v[i][0].y + p1 * v[i][1].y + p2 * v[i][2].y;
        o.z = p0 * v[i][0].z + p1 * v[i][1].z + p2 * v[i][2].z;
        o.w = p0 * v[i][0].w + p1 * v[i][1].w + p2 * v[i][2].w;
        areg[h * 4 + i] = o;
      }
    }
  } else if constexpr (NS0 == 0) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) areg[i] = ld4(src.base + (long long)rows[i] * src.ld + lane * 4);
    gload_first();
  } else {
    // combine: sum of NS0 slabs + bias (+ residual), optional LayerNorm; one wave owns a row (tile32.hpp)
    F4 sl[RPW][NS0], rs[RPW];
    const bool has_res = src.res != nullptr, has_ln = src.gamma != nullptr;
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int s = 0; s < NS0; ++s) sl[i][s] = ld4(src.base + s * src.pstride + (long long)rows[i] * 256 + lane * 4);
    if (has_res) {
#pragma unroll
      for (int i = 0; i < RPW; ++i) rs[i] = ld4(src.res + (long long)rows[i] * src.ldres + lane * 4);
    }
    const F4 bias = ld4(src.bias + lane * 4);
    F4 gm = F4{1.f, 1.f, 1.f, 1.f}, bt = F4{0.f, 0.f, 0.f, 0.f};
    if (has_ln) { gm = ld4(src.gamma + lane * 4); bt = ld4(src.beta + lane * 4); }
    gload_first();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      F4 v = sl[i][0];
#pragma unroll
      for (int s = 1; s < NS0; ++s) v = f4add(v, sl[i][s]);
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
    if (src.out && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < RPW; ++i)
        if (live[i]) st4(src.out + (long long)rows[i] * src.ldout + lane * 4, areg[i]);
    }
  }
  F4 breg[RPW];
  if constexpr (NSRC == 2) {                 // second K segment: plain rows (the stored skip activation)
    const ASrc& s1 = p.src[1];
#pragma unroll
    for (int i = 0; i < RPW; ++i) breg[i] = ld4(s1.base + (long long)rows[i] * s1.ld + lane * 4);
  }
  if constexpr (TRACE) ts[1] = clock_pinned();             // A rows assembled, the first RD weight chunks landed
  constexpr int SEG = PREC == PREC_F32 ? 256 : 128;   // words one 256-wide K segment takes in a row
#pragma unroll
  for (int i = 0; i < RPW; ++i) st_operand<PREC>(As + (wave + i * NW) * ST, lane, areg[i]);
  if constexpr (NSRC == 2) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) st_operand<PREC>(As + (wave + i * NW) * ST + SEG, lane, breg[i]);
  }

  // ---- main loop: one barrier per K chunk; chunk kc multiplies while kc+1 is written to the other LDS buffer and
  //      kc+2..kc+4 are in flight (straight-line after unrolling: KCS is a compile-time constant)
  const int r = lane & 15, g = lane >> 4;
  const int wc = wave & 3, wr = NW == 8 ? (wave >> 2) : 0;      // column tile; first row tile of this wave
  const float* ap = As + (wr * 16 + r) * ST;
  const float* wp = Ws + (wc * 16 + r) * kStripWStride;
  f32x4 acc[CT][RT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  lstore(0);
  if (RD < KCS) gload(RD);
  __syncthreads();
  if constexpr (TRACE) ts[2] = clock_light();               // strip + chunk 0 in LDS, first barrier passed
#pragma unroll
  for (int kc = 0; kc < KCS; ++kc) {
    strip_mma<PREC, RT, CT>(ap, ST, wp + (kc & 1) * BN * kStripWStride, kStripWStride, kc, g, acc);
    if (kc + 1 < KCS) {
      lstore(kc + 1);
      if (kc + 1 + RD < KCS) gload(kc + 1 + RD);
      __syncthreads();
    }
    if constexpr (TRACE) { if (kc == KCS / 2 - 1) ts[3] = clock_light(); }   // half of the K chunks done
  }
  if constexpr (TRACE) {                                    // MFMAs retired
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < RT; ++t) asm volatile("" :: "v"(acc[c][t][0]), "v"(acc[c][t][1]), "v"(acc[c][t][2]), "v"(acc[c][t][3]));
    ts[4] = clock_light();
  }

  // ---- epilogue: 16 lanes write 64 contiguous bytes per row (raw partial slab, or bias + activation)
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int col = n0 + (wc + 4 * c) * 16 + r;
    if (col < p.N) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = m0 + (wr + t) * 16 + g * 4 + i;
          if (row < p.M) {
            if (p.P) {
              p.P[(long long)row * p.N + col] = acc[c][t][i];
            } else {
              float v = acc[c][t][i] + ebias[c];
              if constexpr (ACT == 1) v = gelu_erf(v);
              p.Y[(long long)row * p.ldy + col] = v;
            }
          }
        }
      }
    }
  }
  if constexpr (TRACE) {
    ts[5] = clock_pinned();                 // epilogue stores issued and drained
    if (lane == 0) {
      const long long wg = blockIdx.x + (long long)gridDim.x * blockIdx.y;
      unsigned long long* o = p.trace + (wg * 8 + (wave & 7)) * 8;
      for (int i = 0; i < 6; ++i) o[i] = ts[i];
      o[6] = rt0;
      o[7] = realtime_100mhz();
    }
  }
}

}  // namespace mld