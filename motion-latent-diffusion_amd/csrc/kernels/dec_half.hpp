// Decoder self-attention block with Q | K | V kept as ONE IEEE half per element (round 6; option "dec_half").
//
// The motion-VAE decode (mld_vae.py:186-248; TransformerDecoderLayer.forward_post, cross_attention.py:323-345) runs once behind the reverse
// loop: its arithmetic error is not multiplied by guidance x 50 steps.  tools/precision_attribution_decoder.py -> profiles/r06_decoder_precision.json
// attributes the joint error per GEMM class (both synthetic weight families, the range-contract weight sets): the feed-forward block, the
// out-projection, the skip linears and the final linear need the split-f16 x3 products on heavy-tailed weights (one rounded operand: 1e-3 .. 4e-3
// on the joints), but everything that only feeds the SOFTMAX-weighted average does not --
//   * the in-projection with its input rows rounded to half and the weights kept split (hi + lo): 2 matrix instructions per product instead of 3,
//   * Q | K | V stored as halves: 1.5 KB per frame row instead of 3 KB written and read back (the in-projection is write-heavy, the attention read-heavy),
//   * Q K^T and P V on plain half operands: 1 matrix instruction per product, no hi / lo split of K, V, Q or P anywhere
// together cost <= 2.1e-5 on the joints in every set (contract 1e-3).  finalize's range probe measures this form on the handle's own weights and
// falls back to the fp32-Q|K|V split kernels (attention.hpp, gemm_strip_x3.hpp) above MLDHIP_PROBE_TOL_HALF (include/mldhip.h "Range contract").
//
// Kernels:
//   strip_inproj_h_kernel<RT>   Y[M][768] (halves) = (A[M][256] W^T + bias), columns 0..255 (Q) pre-multiplied by 1/sqrt(64) x log2(e)
//   qkv_to_half_kernel          the same conversion for the [T][768] fp32 projection of decoder layer 0 ("dec_l0_once": one sample's rows)
//   attn_flash_h_kernel         key-blocked masked self-attention over those halves, fp32 softmax / accumulation / output
#pragma once
#include "gemm_strip_x3.hpp"
#include "attention.hpp"

namespace mld {

constexpr float kQScaleLog2 = 0.125f * 1.44269504088896340736f;     // 1/sqrt(head_dim = 64) x log2(e): scores in the log2 domain (softmax on v_exp_f32)

struct InprojHArgs {
  const float* A = nullptr;          // [M][256] fp32 layer input
  const float* W = nullptr;          // fragment-ordered split stream of in_proj_weight (gemm_strip_x3.hpp: 3 pairs x 8 chunks x 2 items)
  const float* bias = nullptr;       // [768]
  unsigned* Y = nullptr;             // [M][384] words = [M][768] halves: q (scaled) | k | v
  int M = 0;
  const int* skip_lens = nullptr; int skip_rpg = 1;      // skip strips made only of padded frames
};

constexpr int kIhXs = 136;           // words per row of the half image / of the output staging tile: 256 halves + 8 words, = 8 mod 16 (conflict-free fragment reads)
template <int RT>
constexpr int inproj_h_lds_bytes() { return 2 * RT * 16 * kIhXs * 4; }      // RT = 4: 69 632 B (two workgroups per CU), RT = 6: 104 448 B

__device__ __forceinline__ float clamp_half_range(float x) { return fminf(fmaxf(x, -65504.f), 65504.f); }

// grid = ceil(M / (16 RT)); block = 512.  Wave w owns output columns 16 w .. 16 w + 15 of every 128-column block and streams their weight
// fragments (hi | lo) register-direct; products TRANSPOSED (weights as the A operand): lane (r, g) ends up with strip row r and four
// consecutive columns 16 w + 4 g .. + 3 -> one 8-byte store of four halves into the staging tile, rows leave as 512-byte runs per column pair.
template <int RT>
__global__ __launch_bounds__(512, RT <= 4 ? 4 : 2) void strip_inproj_h_kernel(InprojHArgs p) {
  constexpr int BM = RT * 16, XS = kIhXs, RING = 4;
#if defined(MLDHIP_SIM)
  unsigned* smem = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_ih[];
  unsigned* smem = smem_ih;
#endif
  unsigned* Xs = smem;               // [BM][136] the strip as halves (A rows rounded once: the "a16" of a16w32)
  unsigned* St = Xs + BM * XS;       // [BM][136] one column pair (256 columns) of the output as halves
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;

  if (p.skip_lens) {                 // uniform exit for strips of padded frames only (gemm_strip_x3.hpp)
    const int t0 = m0, t1 = (t0 + BM < p.M ? t0 + BM : p.M) - 1;
    bool all_padding = true;
    for (int b = t0 / p.skip_rpg; b <= t1 / p.skip_rpg; ++b) {
      const int first = (t0 > b * p.skip_rpg ? t0 : b * p.skip_rpg) - b * p.skip_rpg;
      if (first < p.skip_lens[b]) { all_padding = false; break; }
    }
    if (all_padding) return;
  }

  constexpr int nitems = 3 * 16;
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
  auto mma_item = [&](int j, const U4 (&x)[RT], f32x4 (&acc)[RT]) __attribute__((always_inline)) {
    const int slot = j % RING;
    const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(wl, x[t], acc[t]);
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = mfma_x3_16x16x32(wh, x[t], acc[t]);
    gload(slot);
    sched_fence();
  };

#pragma unroll
  for (int j = 0; j < RING; ++j) gload(j);
  // ---- prologue: the strip's rows -> halves (one wave = one row per pass: 512 contiguous bytes in, 512 out)
#pragma unroll
  for (int j = 0; j < RT * 2; ++j) {
    const int idx = tid + j * 512, row = idx >> 6, c4 = idx & 63;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    const F4 v = ld4_nt(p.A + (size_t)m * 256 + c4 * 4);
    *reinterpret_cast<U2*>(Xs + row * XS + c4 * 2) = U2{split16_hi(clamp_half_range(v.x), clamp_half_range(v.y)), split16_hi(clamp_half_range(v.z), clamp_half_range(v.w))};
  }
  __syncthreads();

  const unsigned* xa = Xs + r * XS + g * 4;
#pragma unroll 1
  for (int pr = 0; pr < 3; ++pr) {
    f32x4 acc0[RT], acc1[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = acc0[t]; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      U4 x[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) x[t] = *reinterpret_cast<const U4*>(xa + t * 16 * XS + 16 * c);
      mma_item(2 * c, x, acc0);
      mma_item(2 * c + 1, x, acc1);
    }
    const F4 bi0 = ld4(p.bias + pr * 256 + wave * 16 + g * 4), bi1 = ld4(p.bias + pr * 256 + 128 + wave * 16 + g * 4);
    const float sc = pr == 0 ? kQScaleLog2 : 1.0f;
    if (pr > 0) __syncthreads();     // the previous pair has left the staging tile
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      unsigned* d = St + (t * 16 + r) * XS + wave * 8 + g * 2;
      *reinterpret_cast<U2*>(d) = U2{split16_hi((acc0[t][0] + bi0.x) * sc, (acc0[t][1] + bi0.y) * sc), split16_hi((acc0[t][2] + bi0.z) * sc, (acc0[t][3] + bi0.w) * sc)};
      *reinterpret_cast<U2*>(d + 64) = U2{split16_hi((acc1[t][0] + bi1.x) * sc, (acc1[t][1] + bi1.y) * sc), split16_hi((acc1[t][2] + bi1.z) * sc, (acc1[t][3] + bi1.w) * sc)};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int idx = tid + j * 512, row = idx >> 5, q = idx & 31;
      if (m0 + row < p.M) {
        const U4 v = *reinterpret_cast<const U4*>(St + row * XS + q * 4);
        st4_nt(reinterpret_cast<float*>(p.Y + (size_t)(m0 + row) * 384 + pr * 128 + q * 4), __builtin_bit_cast(F4, v));
      }
    }
  }
}

// [rows][768] fp32 (q | k | v, bias included) -> halves, q pre-scaled: the projection of decoder layer 0's positional rows ("dec_l0_once": T rows per call)
__global__ __launch_bounds__(256) void qkv_to_half_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int rows) {
  const int n = rows * 96;           // 8 elements per thread
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c8 = i % 96;
    const float sc = c8 < 32 ? kQScaleLog2 : 1.0f;
    const F4 a = ld4(src + (size_t)i * 8), b = ld4(src + (size_t)i * 8 + 4);
    *reinterpret_cast<U4*>(dst + (size_t)i * 4) = U4{split16_hi(a.x * sc, a.y * sc), split16_hi(a.z * sc, a.w * sc), split16_hi(b.x * sc, b.y * sc), split16_hi(b.z * sc, b.w * sc)};
  }
}

// ----------------------------------------------------------------------------------------------
// Key-blocked masked self-attention over half Q | K | V (cross_attention.py:332-333 with the key-padding mask of mld_vae.py:229): the structure of
// attn_flash_x3_kernel (attention.hpp: one workgroup per (sample, head), blocks of 32 keys double buffered in LDS, a wave works on its two query
// tiles w and w + 8 per block, swapped Q K^T so a lane owns one query's scores, lazy reference point, V through ds_read_b64_tr_b16) with
//   * ONE half plane per operand: 16 matrix instructions per block and wave instead of 48, no hi / lo split anywhere (Q arrives pre-scaled and is
//     loaded straight into its fragment registers: lane (r, g)'s eight k-slots are 16 contiguous bytes of the row; K / V blocks are a 16-byte copy
//     per thread into LDS; P is one v_cvt_pk_f16_f32 per pair),
//   * P V TRANSPOSED as well (V^T as the A operand -- the same fragment registers in the other argument): the accumulators of lane (r, g) are query r,
//     head dims 16 dt + 4 g .. + 3, i.e. the lane's OWN query: the running-maximum rescale and the final 1 / l are in-lane multiplies (the x3 kernel
//     fetches them with five lane broadcasts per tile) and the output leaves as 16-byte stores.
// LDS 20 KB per workgroup.  T <= 256 (16 query tiles per (sample, head)).
constexpr int kFlashHStageWords = 2 * 32 * kFlashKStride;      // K plane + V plane of a 32-key block, 40-word rows
constexpr int kFlashHLdsBytes = 2 * kFlashHStageWords * 4;     // 20 480 B

__global__ __launch_bounds__(512, 4) void attn_flash_h_kernel(const unsigned* __restrict__ qkv, float* __restrict__ o,
                                                           const int* __restrict__ lens, int T, int H, int shared_qkv) {
  constexpr int HD = 64, KST = kFlashKStride, NW = 8;
#if defined(MLDHIP_SIM)
  unsigned* smem = reinterpret_cast<unsigned*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) unsigned smem_flash_h[];
  unsigned* smem = smem_flash_h;
#endif
  const int D = H * HD, RW = 3 * D / 2;            // words per packed row
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int bq = shared_qkv ? 0 : b;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int len = lens[b] < T ? lens[b] : T;
  const int nkt = (len + 15) >> 4, nkb = (nkt + 1) >> 1, nqt = nkt;
  const unsigned* base = qkv + (long long)bq * T * RW + h * (HD / 2);

  // ---- staging: threads 0..255 copy K, 256..511 V: key (t & 255) >> 3 of the block, head dims 8 (t & 7) .. + 7 (16 bytes)
  const int skey = (tid & 255) >> 3, part = tid & 7, isv = tid >> 8;
  unsigned kv0, kv1, kv2, kv3;       // (four scalars: a U4 object captured by the two lambdas is kept in scratch by hipcc -- load, drain, spill, reload)
  auto kvload = [&](int kb) {
    const int key = kb * 32 + skey;
    const int kc = key < len ? key : len - 1;      // keys past the length: a valid row's (finite) values; their scores are masked, their P is 0
    const U4 t = *reinterpret_cast<const U4*>(base + (long long)kc * RW + (1 + isv) * (D / 2) + part * 4);
    kv0 = t.x; kv1 = t.y; kv2 = t.z; kv3 = t.w;
  };
  auto kvstore = [&](int kb) {
    *reinterpret_cast<U4*>(smem + (kb & 1) * kFlashHStageWords + isv * 32 * KST + skey * KST + part * 4) = U4{kv0, kv1, kv2, kv3};
  };
  kvload(0);

  bool live[2];
  U4 qf[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = wave + NW * t;
    live[t] = qt < nqt;
    int qrow = qt * 16 + r;
    qrow = qrow < T ? qrow : T - 1;
    const unsigned* qp = base + (long long)qrow * RW + g * 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) qf[t][c] = *reinterpret_cast<const U4*>(qp + c * 16);
  }
  const int nt = live[1] ? 2 : 1;
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  f32x4 oacc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  kvstore(0);
  if (nkb > 1) kvload(1);
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const unsigned* Ks = smem + (kb & 1) * kFlashHStageWords;
    const unsigned* Vs = Ks + 32 * KST;
    if (live[0]) {
      f32x4 s[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) s[t][k2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const U4 kf = *reinterpret_cast<const U4*>(Ks + (k2 * 16 + r) * KST + c * 16 + g * 4);
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (t < nt) s[t][k2] = mfma_x3_16x16x32(kf, qf[t][c], s[t][k2]);      // S^T tile: row = key 4 g + i, column = query r
        }
      U4 pf16[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= nt) continue;
        if (kb * 32 + 32 > len) {                  // the block that crosses the length (wave-uniform test)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[t][k2][i] = kb * 32 + k2 * 16 + g * 4 + i < len ? s[t][k2][i] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3])),
                         fmaxf(fmaxf(s[t][1][0], s[t][1][1]), fmaxf(s[t][1][2], s[t][1][3])));
        mx = max_groups(mx);                       // finite: key 32 kb < len
        if (wave_any(mx > mrun[t] + 8.0f)) {       // lazy reference point (attention.hpp): 2^(s - m) <= 2^8, far inside the half range of P
          const float mnew = fmaxf(mrun[t], mx);
          const float alpha = fast_exp2(mrun[t] - mnew);      // 2^(-inf) = 0 on the first block
          lrun[t] *= alpha;
          mrun[t] = mnew;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) oacc[t][dt][i] *= alpha;      // the lane's own query
        }
        float psum = 0.f;
        float pf[8];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = fast_exp2(s[t][k2][i] - mrun[t]);
            pf[k2 * 4 + i] = e;
            psum += e;
          }
        lrun[t] += sum_groups(psum);
        pf16[t] = U4{split16_hi(pf[0], pf[1]), split16_hi(pf[2], pf[3]), split16_hi(pf[4], pf[5]), split16_hi(pf[6], pf[7])};
      }
      // O^T += V_blk^T P^T: k-slot 8 g + j <-> key (j >> 2) * 16 + 4 g + (j & 3) of the block (the keys whose scores the lane holds)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int tro = (g * 4 + (r >> 2)) * KST + dt * 8 + (r & 3) * 2;
        const U2 a0 = lds_read_tr16_b64(Vs + tro), a1 = lds_read_tr16_b64(Vs + tro + 16 * KST);
        const U4 vf = U4{a0.x, a0.y, a1.x, a1.y};
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (t < nt) oacc[t][dt] = mfma_x3_16x16x32(vf, pf16[t], oacc[t][dt]);    // row = head dim 16 dt + 4 g + i, column = query r
      }
    }
    if (kb + 1 < nkb) {
      kvstore(kb + 1);
      if (kb + 2 < nkb) kvload(kb + 2);
    }
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) continue;
    const int q = (wave + NW * t) * 16 + r;
    if (q < T) {
      const float inv = 1.0f / lrun[t];
      float* op = o + (long long)(b * T + q) * D + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) st4(op + dt * 16, F4{oacc[t][dt][0] * inv, oacc[t][dt][1] * inv, oacc[t][dt][2] * inv, oacc[t][dt][3] * inv});
    }
  }
}

}  // namespace mld
