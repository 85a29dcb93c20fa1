// hipsim -- a tiny functional simulator of the gfx950 execution model, TEST-ONLY.
//
// Purpose: this build container has no GPU and GPU minutes are rationed, so the HIP kernels in
// motion-latent-diffusion_amd/csrc/kernels are compiled a second time for the host with
// -DMLDHIP_SIM and executed by this header: one fiber per work-item, 64-lane wavefronts,
// __syncthreads(), wave shuffles and a bit-exact model of v_mfma_f32_16x16x4_f32 (a k-ordered
// fmaf chain, cdna_hip_programming.md §3).  It lets pytest check every index computation of
// the kernels against the oracle BEFORE the code ever reaches an MI355X.
//
// It is NOT a fallback and NOT part of the product: libmldhip.so is built by hipcc for gfx950
// only, and the mld_hip package never loads the simulator library (tests/hipemu builds
// libmldhip_sim.so into tests/hipemu/ and only tests/ open it).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace hipsim {

struct uint3_ { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct WaveState {
  int alive = 0, count = 0;
  unsigned gen = 0;
  alignas(16) unsigned char buf[2][64][48];   // per-lane operand slots, double buffered
  alignas(16) float res[2][64][4];            // per-lane MFMA results
};

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  uint3_ tid{0, 0, 0};
  int lane = 0, wave = 0;
};

struct BlockState {
  int nthreads = 0, alive = 0, bar_count = 0;
  unsigned bar_gen = 0;
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  uint3_ bid{0, 0, 0};
  dim3 bdim, gdim;
  std::function<void()> body;
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  std::vector<unsigned char> dyn_smem;
};

BlockState& blk();
void yield();
void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
void launch_coresident(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);   // all blocks alive at once (inter-workgroup hand-offs)
void sync_threads();

// ---- wave-level rendezvous: every live lane of the wave must call the same op ----
// Deposits `nbytes` of payload, waits for the whole wave, returns the slot index used.
int wave_arrive(const void* payload, int nbytes, int mfma_kind /*0 none, 1 f32 16x16x4, 2 bf16 16x16x32, 3 fp8 16x16x32, 4 f16 16x16x32*/);

inline float shfl_xor(float v, int mask) {
  BlockState& b = blk();
  int slot = wave_arrive(&v, 4, 0);
  WaveState& w = b.waves[b.cur->wave];
  float r;
  std::memcpy(&r, w.buf[slot][(b.cur->lane ^ mask) & 63], 4);
  return r;
}
inline float shfl(float v, int src) {
  BlockState& b = blk();
  int slot = wave_arrive(&v, 4, 0);
  WaveState& w = b.waves[b.cur->wave];
  float r;
  std::memcpy(&r, w.buf[slot][src & 63], 4);
  return r;
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read, cdna_hip_programming.md "LDS"): inside each 16-lane group the lanes' 8-byte reads
// form a [4][16] block of 16-bit elements -- lane L supplies row L / 4, columns 4 (L % 4) .. + 3 -- and lane i receives COLUMN i of
// it: element j = row j, i.e. element (i % 4) of what lane 4 j + i / 4 fetched.  `own` = the 8 bytes at this lane's own address.
inline void ds_read_tr16_b64(const void* own, unsigned out[2]) {
  BlockState& b = blk();
  int slot = wave_arrive(own, 8, 0);
  WaveState& w = b.waves[b.cur->wave];
  const int lane = b.cur->lane, i = lane & 15, base = lane & ~15;
  unsigned short e[4];
  for (int j = 0; j < 4; ++j) {
    unsigned short h[4];
    std::memcpy(h, w.buf[slot][base + 4 * j + (i >> 2)], 8);
    e[j] = h[i & 3];
  }
  out[0] = (unsigned)e[0] | ((unsigned)e[1] << 16);
  out[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C/D col=l&15,row=(l>>4)*4+reg
inline f32x4 mfma_f32_16x16x4(float a, float bb, f32x4 c) {
  BlockState& b = blk();
  float payload[6] = {a, bb, c[0], c[1], c[2], c[3]};
  int slot = wave_arrive(payload, 24, 1);
  WaveState& w = b.waves[b.cur->wave];
  const float* r = w.res[slot][b.cur->lane];
  f32x4 d = {r[0], r[1], r[2], r[3]};
  return d;
}

// v_mfma_f32_16x16x32_bf16: 8 bf16 of A and of B per lane (k = 8*(l>>4)+j), fp32 accumulate
inline f32x4 mfma_bf16_16x16x32(const unsigned (&a)[4], const unsigned (&bq)[4], f32x4 c) {
  BlockState& b = blk();
  unsigned payload[12];
  for (int i = 0; i < 4; ++i) { payload[i] = a[i]; payload[4 + i] = bq[i]; }
  float cf[4] = {c[0], c[1], c[2], c[3]};
  std::memcpy(&payload[8], cf, 16);
  int slot = wave_arrive(payload, 48, 2);
  WaveState& w = b.waves[b.cur->wave];
  const float* r = w.res[slot][b.cur->lane];
  f32x4 d = {r[0], r[1], r[2], r[3]};
  return d;
}

// v_mfma_f32_16x16x32_f16: same operand layout with IEEE half elements
inline f32x4 mfma_f16_16x16x32(const unsigned (&a)[4], const unsigned (&bq)[4], f32x4 c) {
  BlockState& b = blk();
  unsigned payload[12];
  for (int i = 0; i < 4; ++i) { payload[i] = a[i]; payload[4 + i] = bq[i]; }
  float cf[4] = {c[0], c[1], c[2], c[3]};
  std::memcpy(&payload[8], cf, 16);
  int slot = wave_arrive(payload, 48, 4);
  WaveState& w = b.waves[b.cur->wave];
  const float* r = w.res[slot][b.cur->lane];
  f32x4 d = {r[0], r[1], r[2], r[3]};
  return d;
}

// OCP e4m3fn: 1-4-3, bias 7, max 448, no infinities; round-to-nearest-even (inputs are pre-clamped to +-448)
inline float fp8_e4m3_value(unsigned b) {
  const int sgn = (b >> 7) & 1, ex = (b >> 3) & 15, man = b & 7;
  const float v = ex == 0 ? std::ldexp((float)man, -9) : std::ldexp(1.0f + man / 8.0f, ex - 7);
  return sgn ? -v : v;
}
inline unsigned fp8_e4m3_bits(float x) {
  const unsigned sgn = std::signbit(x) ? 0x80u : 0u;
  float a = std::fabs(x);
  if (!(a > 0.f)) return sgn;
  if (a >= 448.f) return sgn | 0x7Eu;
  int e;
  (void)std::frexp(a, &e);                       // a = m * 2^e, m in [0.5, 1): value exponent e-1
  int ex = e - 1 + 7;
  float q;
  if (ex <= 0) { q = std::nearbyint(std::ldexp(a, 9)); ex = 0; if (q >= 8.f) { ex = 1; q = 0.f; } return sgn | (unsigned)(ex << 3) | (unsigned)q; }
  q = std::nearbyint(std::ldexp(a, 3 - (e - 1))) - 8.f;     // mantissa steps above 1.0 (ties to even under the default mode)
  if (q >= 8.f) { q = 0.f; ++ex; }
  if (ex > 15 || (ex == 15 && q > 6.f)) return sgn | 0x7Eu;
  return sgn | (unsigned)(ex << 3) | (unsigned)q;
}
// v_mfma_f32_16x16x32_fp8_fp8: 8 e4m3 bytes of A and of B per lane (k = 8*(l>>4)+j), fp32 accumulate
inline f32x4 mfma_fp8_16x16x32(const unsigned (&a)[2], const unsigned (&bq)[2], f32x4 c) {
  BlockState& b = blk();
  unsigned payload[12] = {a[0], a[1], 0, 0, bq[0], bq[1], 0, 0, 0, 0, 0, 0};
  float cf[4] = {c[0], c[1], c[2], c[3]};
  std::memcpy(&payload[8], cf, 16);
  int slot = wave_arrive(payload, 48, 3);
  WaveState& w = b.waves[b.cur->wave];
  const float* r = w.res[slot][b.cur->lane];
  f32x4 d = {r[0], r[1], r[2], r[3]};
  return d;
}

}  // namespace hipsim

// ------------------------------------------------------------------ HIP surface used by the kernels
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

#define threadIdx (hipsim::blk().cur->tid)
#define blockIdx (hipsim::blk().bid)
#define blockDim (hipsim::blk().bdim)
#define gridDim (hipsim::blk().gdim)
#define __syncthreads() hipsim::sync_threads()

using hipsim::dim3;
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- the sliver of the HIP runtime the engine uses, on host memory
typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return 0;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hipsim"; }
