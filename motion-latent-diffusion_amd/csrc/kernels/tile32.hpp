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
};

constexpr int kT32Stride = 260;                         // LDS row stride (floats): 256 + 4 pad
constexpr int kT32LdsFloats = (32 + 64) * kT32Stride;   // A tile + W tile
constexpr int kT32LdsBytes = kT32LdsFloats * 4;         // 99,840 B -> one workgroup per CU

// One wave assembles one 256-wide row; lane l owns columns 4l..4l+3.
__device__ __forceinline__ F4 assemble_row(const ASrc& s, int row, int M, int col0, int lane, bool write_back) {
  F4 v;
  if (s.attn_R > 0) {
    // nn.MultiheadAttention over the 3 tokens of one sample (cross_attention.py:265-266):
    // head = lane >> 4 (64 dims = 16 lanes x 4), q pre-scaled by 1/sqrt(64), softmax over 3 keys.
    const int R = s.attn_R;
    const int tok = row / R, smp = row - tok * R;
    const float* q = s.base + (long long)row * 768 + lane * 4;
    const F4 qv = ld4(q);
    F4 kv[3], vv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* kr = s.base + (long long)(j * R + smp) * 768 + 256 + lane * 4;
      kv[j] = ld4(kr);
      vv[j] = ld4(kr + 256);
    }
    float sc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float d = qv.x * kv[j].x;
      d = fmaf(qv.y, kv[j].y, d);
      d = fmaf(qv.z, kv[j].z, d);
      d = fmaf(qv.w, kv[j].w, d);
      sc[j] = sum16(d) * 0.125f;
    }
    const float m = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
    const float e0 = expf(sc[0] - m), e1 = expf(sc[1] - m), e2 = expf(sc[2] - m);
    const float inv = 1.0f / (e0 + e1 + e2);
    const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
    v.x = p0 * vv[0].x + p1 * vv[1].x + p2 * vv[2].x;
    v.y = p0 * vv[0].y + p1 * vv[1].y + p2 * vv[2].y;
    v.z = p0 * vv[0].z + p1 * vv[1].z + p2 * vv[2].z;
    v.w = p0 * vv[0].w + p1 * vv[1].w + p2 * vv[2].w;
    return v;
  }
  if (s.nsplit == 0) return ld4(s.base + (long long)row * s.ld + col0 + lane * 4);
  // ---- combine: sum of slabs + bias + residual, then optional LayerNorm (all in this wave)
  v = ld4(s.base + (long long)row * 256 + lane * 4);
  for (int z = 1; z < s.nsplit; ++z) {
    const F4 t = ld4(s.base + z * s.pstride + (long long)row * 256 + lane * 4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (s.bias) {
    const F4 t = ld4(s.bias + lane * 4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (s.res) {
    const F4 t = ld4(s.res + (long long)row * s.ldres + lane * 4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (s.gamma) {
    const float mean = sum64(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float var = sum64(a * a + b * b + c * c + d * d) * (1.0f / 256.0f);
    const float rs = rsqrtf(var + kLnEps);
    const F4 gm = ld4(s.gamma + lane * 4), bt = ld4(s.beta + lane * 4);
    v = F4{a * rs * gm.x + bt.x, b * rs * gm.y + bt.y, c * rs * gm.z + bt.z, d * rs * gm.w + bt.w};
  }
  if (write_back && s.out) st4(s.out + (long long)row * s.ldout + lane * 4, v);
  return v;
}

// grid = (ceil(M/32), N/64, K/256); block = 512 (8 waves: wave w -> column tile w&3, row tile w>>2).
__global__ __launch_bounds__(512) void gemm_tile32_kernel(Tile32Args p) {
#if defined(MLDHIP_SIM)
  float* smem = reinterpret_cast<float*>(hipsim::blk().dyn_smem.data());
#else
  extern __shared__ __attribute__((aligned(16))) float smem[];
#endif
  float* As = smem;                          // [32][260]
  float* Ws = smem + 32 * kT32Stride;        // [64][260]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64, z = blockIdx.z;
  const bool second = z >= p.nz0;
  const ASrc& src = second ? p.src[1] : p.src[0];
  const int acol = (second ? z - p.nz0 : z) * 256;
  const int wcol = z * 256;

  // ---- issue everything: 8 weight rows + 4 A rows per wave, one coalesced 1 KiB row per instruction
  F4 wreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int n = n0 + wave + i * 8;
    n = n < p.N ? n : p.N - 1;
    wreg[i] = ld4(p.W + (long long)n * p.ldw + wcol + lane * 4);
  }
  F4 areg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = m0 + wave + i * 8;
    const bool live = row < p.M;
    row = live ? row : p.M - 1;
    areg[i] = assemble_row(src, row, p.M, acol, lane, live && blockIdx.y == 0);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) st4(Ws + (wave + i * 8) * kT32Stride + lane * 4, wreg[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) st4(As + (wave + i * 8) * kT32Stride + lane * 4, areg[i]);
  __syncthreads();

  // ---- 64 MFMAs per wave: one 16x16 tile over K = 256, two accumulators to hide the MFMA latency
  const int r = lane & 15, g = lane >> 4;
  const int ct = wave & 3, rt = wave >> 2;
  const float* ap = As + (rt * 16 + r) * kT32Stride + g * 8;
  const float* wp = Ws + (ct * 16 + r) * kT32Stride + g * 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
    const F4 a0 = ld4(ap + kc * 32), a1 = ld4(ap + kc * 32 + 4);
    const F4 b0 = ld4(wp + kc * 32), b1 = ld4(wp + kc * 32 + 4);
    acc0 = mfma_f32_16x16x4(a0.x, b0.x, acc0);
    acc1 = mfma_f32_16x16x4(a0.y, b0.y, acc1);
    acc0 = mfma_f32_16x16x4(a0.z, b0.z, acc0);
    acc1 = mfma_f32_16x16x4(a0.w, b0.w, acc1);
    acc0 = mfma_f32_16x16x4(a1.x, b1.x, acc0);
    acc1 = mfma_f32_16x16x4(a1.y, b1.y, acc1);
    acc0 = mfma_f32_16x16x4(a1.z, b1.z, acc0);
    acc1 = mfma_f32_16x16x4(a1.w, b1.w, acc1);
  }
  const f32x4 acc = acc0 + acc1;
  const int col = n0 + ct * 16 + r;
  if (col >= p.N) return;
  if (p.P) {
    float* P = p.P + z * p.pstride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + rt * 16 + g * 4 + i;
      if (row < p.M) P[(long long)row * p.N + col] = acc[i];
    }
  } else {
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + rt * 16 + g * 4 + i;
      if (row < p.M) {
        float v = acc[i] + bv;
        if (p.act == 1) v = gelu_erf(v);
        else if (p.act == 2) v = silu(v);
        p.Y[(long long)row * p.ldy + col] = v;
      }
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
