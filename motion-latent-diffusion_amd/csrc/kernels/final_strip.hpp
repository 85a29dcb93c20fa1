// The end of MldVae.decode as ONE row-strip launch (split-f16 operands, weights register-direct):
//   feats[M, NF] = mask( LayerNorm(X; gamma, beta) W^T + bias )          (mld_vae.py:240-245: decoder.norm, final_layer, output[~mask.T] = 0)
// for D = 256 and 256 < NF <= 264 (HumanML3D: 263; the bound is the parking space: 48 x NF results in the strip's 48 x 264-word image).  Replaces layernorm_rows_kernel + the staged K = 256 GEMM, whose 64 x 128 tiles re-read the
// normalised rows once per 128-column tile (N = 263 -> three tiles: 1.2 GB instead of 0.4 GB at 2 048 motions, the third tile for seven
// columns) after a 0.4 GB write + read of the normalised tensor itself, and store 263-float rows with 4-byte stores.  Here a workgroup
// owns 48 rows: it normalises them while it loads them (one wave per row: the same arithmetic, in the same order, as
// layernorm_rows_kernel), keeps them in LDS as a split image, multiplies them with all three 128-column blocks of the zero-padded weight
// (wave w owns columns 16 w .. 16 w + 15 of each block; items [chunk][block] from a fragment-ordered stream, the A fragments of a
// chunk feed three items), parks the 48 x NF results in the image's own rows and writes them out as ONE contiguous block of 48 NF
// floats with 16-byte stores (rows of the output are NF floats apart: a strip of rows is contiguous; 48 NF x 4 bytes is a multiple
// of 16 for every NF).  HBM traffic: the input once, the output once.
#pragma once
#include "gemm_strip_x3.hpp"

namespace mld {

struct FinalStripArgs {
  const float* X = nullptr;          // [M][256] decoder output before decoder.norm
  const float* gamma = nullptr; const float* beta = nullptr;     // decoder.norm
  const float* W = nullptr;          // fragment-ordered stream of the weight padded to 384 rows: 8 chunks x [block 0, block 1, block 2]
  const float* bias = nullptr;       // [NF]
  float* Y = nullptr;                // [M][NF]
  int M = 0, NF = 0;
  const int* lens = nullptr; int rpg = 1;      // rows with (row % rpg) >= lens[row / rpg] are written as zeros
};

constexpr int kFinalStripRows = 48, kFinalStripItems = 24;
constexpr int final_strip_lds_bytes() { return (kFinalStripRows * kFsXs + kFinalStripRows) * 4; }

// finalize-time: a weight with fewer than 128 valid rows in its last block -> fragment-ordered items, missing rows as zeros
// (pack_loop_stream_kernel<true> with a row limit: LoopItem.pad = valid rows of the item's 128-row block).  grid = items, block = 512.
__global__ __launch_bounds__(512) void pack_stream_rows_kernel(const float* __restrict__ arena, const LoopItem* __restrict__ items,
                                                               float* __restrict__ out) {
  const LoopItem it = items[blockIdx.x];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const bool live = 16 * w + r < it.pad;
  const float* src = arena + it.src + (long long)(live ? 16 * w + r : 0) * it.ld;
  float* dst = out + (long long)blockIdx.x * kLoopItemFloats + threadIdx.x * 8;
  const float m = live ? 1.f : 0.f;
  const F4 a = ld4(src + 8 * g), b = ld4(src + 8 * g + 4);
  U4 hi, lo;
  split16_pair(a.x * m, a.y * m, hi.x, lo.x);
  split16_pair(a.z * m, a.w * m, hi.y, lo.y);
  split16_pair(b.x * m, b.y * m, hi.z, lo.z);
  split16_pair(b.z * m, b.w * m, hi.w, lo.w);
  *reinterpret_cast<U4*>(dst) = hi;
  *reinterpret_cast<U4*>(dst + 4) = lo;
}

// grid = ceil(M / 48); block = 512
__global__ __launch_bounds__(512, 4) void final_strip_x3_kernel(FinalStripArgs p) {
  constexpr int RT = 3, BM = RT * 16, XS = kFsXs;
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem_fin[];
  float* smem = smem_fin;
#endif
  float* Xs = smem;                                        // [48][264]: the split image of the normalised strip, then the fp32 results
  float* rmask = Xs + BM * XS;                             // [48] 1 / 0: row is a valid frame
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int col0 = wave * 16 + r;

  const float* gsrc = p.W + tid * 8;
  constexpr int RING = 4;
  F4 ring[RING][2];
  int gitem = 0;
  auto gload = [&](int slot) __attribute__((always_inline)) {
    const int it = gitem < kFinalStripItems ? gitem : kFinalStripItems - 1;      // past the end: a redundant load, never multiplied
    const float* s = gsrc + (unsigned)it * (unsigned)kLoopItemFloats;
    ring[slot][0] = ld4(s);
    ring[slot][1] = ld4(s + 4);
    ++gitem;
  };
  // ---- prologue: LayerNorm of the strip's rows while they are loaded (one wave per row and pass, 4 columns per lane), -> split image.
  // Issue order: the six rows of this wave (HBM: the longest wait) first, then the first items of the weight stream (L2), the
  // LayerNorm parameters and the lengths; nothing is reduced before everything is in flight (vmcnt is an in-order counter).
  {
    F4 xr[RT * 2];
#pragma unroll
    for (int j = 0; j < RT * 2; ++j) {
      int m = m0 + wave + 8 * j;
      m = m < p.M ? m : p.M - 1;
      xr[j] = ld4(p.X + (size_t)m * 256 + lane * 4);
    }
    sched_fence();
#pragma unroll
    for (int j = 0; j < RING; ++j) gload(j);
    const F4 gm = ld4(p.gamma + lane * 4), bt = ld4(p.beta + lane * 4);
    sched_fence();
    if (tid < BM) {
      const int m = m0 + tid < p.M ? m0 + tid : p.M - 1;
      float v = m0 + tid < p.M ? 1.f : 0.f;
      if (p.lens) {
        const int grp = m / p.rpg;
        v = (m - grp * p.rpg) < p.lens[grp] ? v : 0.f;
      }
      rmask[tid] = v;
    }
#pragma unroll
    for (int j = 0; j < RT * 2; ++j) {
      const int row = wave + 8 * j;                      // (tid + 512 j) >> 6
      const F4 x = xr[j];
      const float mean = sum64(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
      const float a = x.x - mean, b = x.y - mean, c = x.z - mean, d = x.w - mean;
      const float var = sum64(a * a + b * b + c * c + d * d) * (1.0f / 256.0f);
      const float rs = rsqrtf(var + kLnEps);
      unsigned h0, l0, h1, l1;
      split16_pair(a * rs * gm.x + bt.x, b * rs * gm.y + bt.y, h0, l0);
      split16_pair(c * rs * gm.z + bt.z, d * rs * gm.w + bt.w, h1, l1);
      unsigned* dd = reinterpret_cast<unsigned*>(Xs) + row * XS + (lane >> 3) * 32 + (lane & 7) * 2;
      *reinterpret_cast<U2*>(dd) = U2{h0, h1};
      *reinterpret_cast<U2*>(dd + 16) = U2{l0, l1};
    }
  }
  __syncthreads();

  // ---- products: 8 chunks x 3 column blocks
  f32x4 acc[3][RT];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xa = Xs + r * XS + g * 4;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    F4 x[RT][2];
#pragma unroll
    for (int t = 0; t < RT; ++t) { x[t][0] = ld4(xa + t * 16 * XS + 32 * c); x[t][1] = ld4(xa + t * 16 * XS + 32 * c + 16); }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int slot = (3 * c + b) % RING;
      const U4 wh = __builtin_bit_cast(U4, ring[slot][0]), wl = __builtin_bit_cast(U4, ring[slot][1]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[b][t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][1]), wh, acc[b][t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[b][t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wl, acc[b][t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[b][t] = mfma_x3_16x16x32(__builtin_bit_cast(U4, x[t][0]), wh, acc[b][t]);
      gload(slot);
      sched_fence();
    }
  }
  __syncthreads();                                         // every wave is done with the image: its rows now take the results

  // ---- bias, padded-frame zeroing, results parked row-major [48][NF] PACKED (row stride NF): the strip's output block as it lies in memory
  float* Out = Xs;                                         // 48 NF <= 48 x 264 floats (the engine builds the stream only for NF <= 264)
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int col = b * 128 + col0;
    const float bi = p.bias[col < p.NF ? col : p.NF - 1];
    if (col < p.NF) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = t * 16 + g * 4 + i;
          Out[row * p.NF + col] = (acc[b][t][i] + bi) * rmask[row];
        }
    }
  }
  __syncthreads();
  {
    const int rows = m0 + BM <= p.M ? BM : p.M - m0;
    const int n = rows * p.NF;                             // floats of the block; its first byte is 16-byte aligned (48 NF x 4 per strip)
    float* dst = p.Y + (size_t)m0 * p.NF;
    if ((reinterpret_cast<unsigned long long>(p.Y) & 15) == 0) {      // (a caller's buffer promises float alignment only: gemm.hpp store_tile_from_lds)
      for (int q = tid * 4; q + 3 < n; q += 512 * 4) st4(dst + q, ld4(Out + q));
      if (tid < (n & 3)) dst[(n & ~3) + tid] = Out[(n & ~3) + tid];    // a partial last strip can end off a 16-byte boundary
    } else {
      for (int q = tid; q < n; q += 512) dst[q] = Out[q];
    }
  }
}

}  // namespace mld
