// Small fused row kernels of the MLD engine: token assembly, final-norm + CFG + DDIM update,
// LayerNorm rows, decoder query init, feats -> joints.  All are latency/HBM-trivial (KBs per call);
// they exist to keep the 50-step loop free of host work so it can live in one hipGraph.
#pragma once
#include "rt.hpp"

namespace mld {

// block-wide sum for a 256-thread workgroup (4 waves); every thread gets the result.
__device__ __forceinline__ float block_sum_256(float v, float* sh /*[4]*/, int tid) {
  v = sum64(v);
  __syncthreads();                 // protect sh from the previous use
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ----------------------------------------------------------------------------------------------
// Token assembly at the start of the reverse process for one chain of `Bc` motions starting at b0
// (mld_denoiser.py:143-196: tokens [latent, time, text]; rows token-major s*Rc + r with Rc = 2*Bc and the
// unconditional half first, mld.py:224-231,325):
//   X0[r]          = init_latents[b] * init_sigma + pe[0]        (both CFG halves share the latent)
//   X0[Rc + r]     = T1[step 0]                                   (time MLP + pe[1], precomputed)
//   X0[2Rc + r]    = TP[b0 + b] (uncond) / TP[B + b0 + b] (cond)  (text projection + pe[2])
// grid = Bc, block = 256 (= latent width).
__global__ __launch_bounds__(256) void init_chain_kernel(const float* __restrict__ init_lat, float* __restrict__ lat,
                                                         float* __restrict__ X0, const float* __restrict__ pe0,
                                                         const float* __restrict__ t1_row, const float* __restrict__ TP,
                                                         int B, int b0, int Bc, float init_sigma) {
  const int b = blockIdx.x, d = threadIdx.x, Rc = 2 * Bc;
  const float x = init_lat[(long long)b * 256 + d] * init_sigma;
  lat[(long long)b * 256 + d] = x;
  const float tok = x + pe0[d];
  X0[(long long)b * 256 + d] = tok;
  X0[(long long)(Bc + b) * 256 + d] = tok;
  const float tt = t1_row[d];
  X0[(long long)(Rc + b) * 256 + d] = tt;
  X0[(long long)(Rc + Bc + b) * 256 + d] = tt;
  X0[(long long)(2 * Rc + b) * 256 + d] = TP[(long long)(b0 + b) * 256 + d];
  X0[(long long)(2 * Rc + Bc + b) * 256 + d] = TP[(long long)(B + b0 + b) * 256 + d];
}

// Split-bf16 image of the weight arena (finalize-time; precision modes that run staged GEMMs on split-bf16 MFMAs): every aligned
// group of 32 floats -- one 32-wide K chunk of one weight row, tensors start on 64-float boundaries and the staged GEMMs take K % 32 == 0
// -- becomes the 32 words a staged GEMM keeps in LDS for it: 16 words of bf16 high parts (element 2w in the low half of word w), then
// 16 words of bf16 low parts (rt.hpp split_bf16_pair; gemm.hpp lstore).  Same size and addressing as the fp32 arena, so a GEMM
// reads W from it at the same offset and stores the 16-byte pieces to LDS untouched.  grid = ceil(groups / 16), block = 256.
__global__ __launch_bounds__(256) void split_bf16_weights_kernel(const float* __restrict__ w, float* __restrict__ out, long long groups) {
  const long long grp = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int pr = threadIdx.x & 15;                      // pair (2 pr, 2 pr + 1) of the group
  if (grp >= groups) return;
  const float a = w[grp * 32 + 2 * pr], b = w[grp * 32 + 2 * pr + 1];
  unsigned hi, lo;
  split16_pair(a, b, hi, lo);
  unsigned* o = reinterpret_cast<unsigned*>(out) + grp * 32;
  o[pr] = hi;
  o[16 + pr] = lo;
}

struct DdimCoef { float sqrt_at, sqrt_1mat, sqrt_ap, sqrt_1map; };   // DDIM eta=0 coefficients of one step

// LayerNorm over rows of width 256: one wave per row, 4 rows per workgroup.
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row = blockIdx.x * 4 + wave;
  const bool live = row < M;
  row = live ? row : M - 1;
  const F4 x = ld4(X + (long long)row * 256 + lane * 4);
  const float mean = sum64(x.x + x.y + x.z + x.w) * (1.0f / 256.0f);
  const float a = x.x - mean, b = x.y - mean, c = x.z - mean, d = x.w - mean;
  const float var = sum64(a * a + b * b + c * c + d * d) * (1.0f / 256.0f);
  const float rs = rsqrtf(var + kLnEps);
  const F4 gm = ld4(gamma + lane * 4), bt = ld4(beta + lane * 4);
  if (live) st4(Y + (long long)row * 256 + lane * 4, F4{a * rs * gm.x + bt.x, b * rs * gm.y + bt.y, c * rs * gm.z + bt.z, d * rs * gm.w + bt.w});
}

// Decoder queries: zeros + learned PE (mld_vae.py:190,224): Hq[b*T + t] = pe[t].  grid = B*T/… rows.
__global__ __launch_bounds__(256) void init_queries_kernel(float* __restrict__ Hq, const float* __restrict__ pe,
                                                           int B, int T, int D) {
  const long long n4 = (long long)B * T * D / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int d = int(e % D);
    const int t = int((e / D) % T);
    st4(Hq + e, ld4(pe + (long long)t * D + d));
  }
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ a,
                                                       const float* __restrict__ vec, int rows, int D) {
  // dst[r][d] = a[r][d] + vec[d]
  const long long n = (long long)rows * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = a[i] + vec[i % D];
}

// ---- mldhip_sample_many: the requests' inputs into the engine's batch buffers and the results back out, ONE launch each way.  (As a
// hipMemcpyAsync per tensor a 32-request call issued 216 copies of ~4.4 us each, back to back on the stream: 0.95 ms of a 51 ms call.)
// Pointer tables travel as kernel arguments (<= 64 requests).
constexpr int kMaxRequests = 64;
struct GatherArgs {
  const float* text[kMaxRequests];     // [2 b][TD] per request (unconditional half first) or NULL (action condition)
  const float* lat[kMaxRequests];      // [b][D]
  int off[kMaxRequests], nb[kMaxRequests];
  float* text_in; float* lat_in;       // [2 Btot][TD], [Btot][D]
  int Btot, TD, D;
};
// grid = (requests, chunks), block = 256: request blockIdx.x, elements strided over blockIdx.y (4-byte accesses: caller pointers carry
// no alignment promise beyond float)
__global__ __launch_bounds__(256) void gather_requests_kernel(GatherArgs a) {
  const int i = blockIdx.x, b = a.nb[i], o = a.off[i];
  const long long nt = a.text[i] ? (long long)b * a.TD : 0, nl = (long long)b * a.D;
  const float* t = a.text[i];
  const float* l = a.lat[i];
  float* tu = a.text_in + (long long)o * a.TD;
  float* tc = a.text_in + ((long long)a.Btot + o) * a.TD;
  float* ld = a.lat_in + (long long)o * a.D;
  for (long long k = (long long)blockIdx.y * 256 + threadIdx.x; k < 2 * nt + nl; k += (long long)gridDim.y * 256) {
    if (k < nt) tu[k] = t[k];
    else if (k < 2 * nt) tc[k - nt] = t[k];
    else ld[k - 2 * nt] = l[k - 2 * nt];
  }
}
struct ScatterArgs {
  float* lat_out[kMaxRequests];        // [b][D] or NULL
  float* feats_out[kMaxRequests];      // [b][tmax][NF] or NULL
  float* joints_out[kMaxRequests];     // [b][tmax][NJ] or NULL
  int off[kMaxRequests], nb[kMaxRequests], tmax[kMaxRequests];
  const float* lat; const float* feats; const float* joints;      // engine-side: [Btot][D], [Btot][T][NF], [Btot][T][NJ]
  int T, D, NF, NJ;
};
// grid = (requests, motions of the largest request), block = 256: motion blockIdx.y of request blockIdx.x (rows of a motion are
// contiguous on both sides: tmax x width floats; 4-byte accesses -- 66-float joint rows are not 16-byte aligned in general)
__global__ __launch_bounds__(256) void scatter_results_kernel(ScatterArgs a) {
  const int i = blockIdx.x, k = blockIdx.y;
  if (k >= a.nb[i]) return;
  const long long m = a.off[i] + k, ti = a.tmax[i];
  if (a.lat_out[i])
    for (int d = threadIdx.x; d < a.D; d += 256) a.lat_out[i][(long long)k * a.D + d] = a.lat[m * a.D + d];
  if (a.feats_out[i]) {
    const float* s = a.feats + m * a.T * a.NF;
    float* d = a.feats_out[i] + (long long)k * ti * a.NF;
    for (long long q = threadIdx.x; q < ti * a.NF; q += 256) d[q] = s[q];
  }
  if (a.joints_out[i]) {
    const float* s = a.joints + m * a.T * a.NJ;
    float* d = a.joints_out[i] + (long long)k * ti * a.NJ;
    for (long long q = threadIdx.x; q < ti * a.NJ; q += 256) d[q] = s[q];
  }
}

// EmbedAction in eval mode (mld_denoiser.py:249-260) + the token-2 positional row:
//   dst[r] = (r < nuncond ? 0 : table[labels[r]]) + pe2        grid = rows, block = 256 (= latent width)
__global__ __launch_bounds__(256) void action_rows_kernel(float* __restrict__ dst, const float* __restrict__ table,
                                                          const float* __restrict__ pe2, const int* __restrict__ labels,
                                                          int nuncond) {
  const int r = blockIdx.x, d = threadIdx.x;
  float v = pe2[d];
  if (r >= nuncond) v = table[(long long)labels[r] * 256 + d] + v;
  dst[(long long)r * 256 + d] = v;
}

__global__ __launch_bounds__(256) void bcast_rows_kernel(float* __restrict__ dst, const float* __restrict__ vec, int rows, int D) {
  const long long n = (long long)rows * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = vec[i % D];
}

// dst[r][0..KP) = src[r][0..K) followed by zeros (K = 263 features -> KP = 288 so MFMA K chunks stay full)
__global__ __launch_bounds__(256) void pad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int K, int KP) {
  const long long n = (long long)rows * KP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / KP;
    const int c = int(i - r * KP);
    dst[i] = c < K ? src[r * K + c] : 0.f;
  }
}

// Encoder token assembly (mld_vae.py:143-160): X[b*(T+2) + s] = (s < 2 ? global_motion_token[s] : emb[b*T + s-2]) + pe[s]
__global__ __launch_bounds__(256) void enc_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ tok,
                                                         const float* __restrict__ pe, float* __restrict__ X, int B, int T, int D) {
  const int S = T + 2;
  const long long n4 = (long long)B * S * D / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int d = int(e % D);
    const long long row = e / D;
    const int s = int(row % S), b = int(row / S);
    const F4 a = s < 2 ? ld4(tok + (long long)s * D + d) : ld4(emb + ((long long)b * T + s - 2) * D + d);
    const F4 p = ld4(pe + (long long)s * D + d);
    st4(X + e, F4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w});
  }
}

// Encoder head (mld_vae.py:176-183): mu / logvar = final-norm of token rows 0 / 1 of each sample; the
// reparameterised draw uses an injected N(0,1) tensor: latent = mu + exp(logvar)^0.5 * eps.  grid = B, block = 256.
__global__ __launch_bounds__(256) void enc_finish_kernel(const float* __restrict__ H, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ eps,
                                                         float* __restrict__ latent, float* __restrict__ mu, float* __restrict__ logvar, int S) {
  __shared__ float sh[4];
  const int b = blockIdx.x, d = threadIdx.x;
  float out[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float x = H[((long long)b * S + s) * 256 + d];
    if (!gamma) { out[s] = x; continue; }     // ActorAgnosticEncoder has no final norm (actor_vae.py:166-170); uniform branch
    const float mean = block_sum_256(x, sh, d) * (1.0f / 256.0f);
    const float xc = x - mean;
    const float var = block_sum_256(xc * xc, sh, d) * (1.0f / 256.0f);
    out[s] = xc * rsqrtf(var + kLnEps) * gamma[d] + beta[d];
  }
  mu[(long long)b * 256 + d] = out[0];
  logvar[(long long)b * 256 + d] = out[1];
  if (eps) latent[(long long)b * 256 + d] = out[0] + sqrtf(expf(out[1])) * eps[(long long)b * 256 + d];
}

// ----------------------------------------------------------------------------------------------
// feats -> joints: HumanML3DDataModule.feats2joints + recover_from_ric
// (HumanML3D.py:41-45, motion_process.py:362-381,415-432, quaternion.py:16-20,54-73).
// One workgroup per sample.  Wave 0 integrates the root: yaw_t = sum_{s<t} f[s,0] and
// root_xz(t) = sum_{s<=t} R_y(yaw_s)^-1 [f[s-1,1], 0, f[s-1,2]] as two wave-level inclusive scans
// (each lane owns a run of consecutive frames), then all 256 threads rotate the 21 root-relative
// joints of every frame.  Only feature columns 0..66 are read.
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = wave_bcast(v, lane >= off ? lane - off : lane);
    if (lane >= off) v += u;
  }
  return v;
}

// Run-time part of the F16X3 range contract (include/mldhip.h): counts the non-finite elements of a result buffer into a sticky
// device counter.  An operand that left the half range inside a split-f16 kernel became inf, then NaN, and reached every later
// value of its motion: the latents after the loop and the joints after the decode are the two places worth looking at.
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ counter) {
  float bad = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned u = __builtin_bit_cast(unsigned, x[i]);
    bad += ((u & 0x7F800000u) == 0x7F800000u) ? 1.f : 0.f;        // exponent all ones: inf or NaN
  }
  bad = sum64(bad);
  if ((threadIdx.x & 63) == 0 && bad > 0.f) {
#if defined(MLDHIP_SIM)
    *counter += (unsigned)bad;
#else
    atomicAdd(counter, (unsigned)bad);
#endif
  }
}

template <int MAXT>
__global__ __launch_bounds__(256) void feats2joints_kernel(const float* __restrict__ feats, float* __restrict__ joints,
                                                           const float* __restrict__ mean, const float* __restrict__ stdv,
                                                           int T, int nfeats, int njoints) {
  __shared__ float s_cos[MAXT], s_sin[MAXT], s_rx[MAXT], s_rz[MAXT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* F = feats + (long long)b * T * nfeats;
  constexpr int PER = (MAXT + 63) / 64;
  if (tid < 64) {
    const float m0 = mean[0], s0 = stdv[0], m1 = mean[1], s1 = stdv[1], m2 = mean[2], s2 = stdv[2];
    // ---- yaw: exclusive prefix sum of the de-normalised rotation velocity
    float loc[PER];
    float run = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = lane * PER + j;            // frame t receives rot_vel of frame t-1
      const float v = (t >= 1 && t < T) ? (F[(long long)(t - 1) * nfeats] * s0 + m0) : 0.f;
      run += v;
      loc[j] = run;
    }
    float incl = wave_incl_scan(run, lane);
    float base = incl - run;
    float cs[PER], sn[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = lane * PER + j;
      const float ang = base + loc[j];
      cs[j] = cosf(ang);
      sn[j] = sinf(ang);
      if (t < T) { s_cos[t] = cs[j]; s_sin[t] = sn[j]; }
    }
    // ---- root xz: inclusive prefix sum of the rotated previous-frame velocity
    float lx[PER], lz[PER];
    float runx = 0.f, runz = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = lane * PER + j;
      float vx = 0.f, vz = 0.f;
      if (t >= 1 && t < T) {
        vx = F[(long long)(t - 1) * nfeats + 1] * s1 + m1;
        vz = F[(long long)(t - 1) * nfeats + 2] * s2 + m2;
      }
      // qrot(qinv((c,0,s,0)), (vx,0,vz)) with u = (0,-s,0), w = c
      const float c = cs[j], s = sn[j];
      const float uvx = -s * vz, uvz = s * vx;
      const float uuvx = -s * uvz, uuvz = s * uvx;
      runx += vx + 2.f * (c * uvx + uuvx);
      runz += vz + 2.f * (c * uvz + uuvz);
      lx[j] = runx;
      lz[j] = runz;
    }
    const float ix = wave_incl_scan(runx, lane), iz = wave_incl_scan(runz, lane);
    const float bx = ix - runx, bz = iz - runz;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int t = lane * PER + j;
      if (t < T) { s_rx[t] = bx + lx[j]; s_rz[t] = bz + lz[j]; }
    }
  }
  __syncthreads();
  const int total = T * njoints;
  for (int idx = tid; idx < total; idx += 256) {
    const int t = idx / njoints, j = idx - t * njoints;
    float* out = joints + ((long long)(b * T + t) * njoints + j) * 3;
    const float rx = s_rx[t], rz = s_rz[t];
    if (j == 0) {
      out[0] = rx;
      out[1] = F[(long long)t * nfeats + 3] * stdv[3] + mean[3];
      out[2] = rz;
    } else {
      const int c0 = 4 + (j - 1) * 3;
      const float px = F[(long long)t * nfeats + c0] * stdv[c0] + mean[c0];
      const float py = F[(long long)t * nfeats + c0 + 1] * stdv[c0 + 1] + mean[c0 + 1];
      const float pz = F[(long long)t * nfeats + c0 + 2] * stdv[c0 + 2] + mean[c0 + 2];
      const float c = s_cos[t], s = s_sin[t];
      const float uvx = -s * pz, uvz = s * px;
      const float uuvx = -s * uvz, uuvz = s * uvx;
      out[0] = px + 2.f * (c * uvx + uuvx) + rx;
      out[1] = py;
      out[2] = pz + 2.f * (c * uvz + uuvz) + rz;
    }
  }
}

}  // namespace mld
