// Device-code prelude for the mldhip kernels (gfx950 / CDNA4 only).
//
// The one build of record is hipcc --offload-arch=gfx950.  The MLDHIP_SIM branch is a test
// seam, not a second backend: tests/hipemu compiles the very same kernel sources for the host
// against a functional model of wave64 + v_mfma_f32_16x16x4_f32 so index math can be checked
// without a GPU.  Nothing in the shipped library or the mld_hip package references it.
#pragma once

#if defined(MLDHIP_SIM)
#include "hipsim.h"
#define MLD_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipsim::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define MLD_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#endif

#include <cstdint>

namespace mld {

constexpr int kWave = 64;   // CDNA wavefront

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32 -- exact-fp32 matrix FMA (bitwise a k-ordered fmaf chain), 157 TF peak.
// Lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15];
// result register r of lane l is D[row = (l>>4)*4 + r][col = l&15].
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
#if defined(MLDHIP_SIM)
  return hipsim::mfma_f32_16x16x4(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float wave_xor(float v, int mask) {
#if defined(MLDHIP_SIM)
  return hipsim::shfl_xor(v, mask);
#else
  return __shfl_xor(v, mask, kWave);
#endif
}

__device__ __forceinline__ float wave_bcast(float v, int src_lane) {
#if defined(MLDHIP_SIM)
  return hipsim::shfl(v, src_lane);
#else
  return __shfl(v, src_lane, kWave);
#endif
}

// sum / max over the 16 lanes that share (lane >> 4): the column axis of an MFMA C tile
__device__ __forceinline__ float sum16(float v) {
  v += wave_xor(v, 1);
  v += wave_xor(v, 2);
  v += wave_xor(v, 4);
  v += wave_xor(v, 8);
  return v;
}
// sum / max over the 4 lanes {l, l^16, l^32, l^48}: the row-group axis of an MFMA C tile
__device__ __forceinline__ float sum_groups(float v) {
  v += wave_xor(v, 16);
  v += wave_xor(v, 32);
  return v;
}
__device__ __forceinline__ float max_groups(float v) {
  v = fmaxf(v, wave_xor(v, 16));
  v = fmaxf(v, wave_xor(v, 32));
  return v;
}
__device__ __forceinline__ float sum64(float v) { return sum_groups(sum16(v)); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

struct alignas(16) F4 { float x, y, z, w; };
__device__ __forceinline__ F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }
__device__ __forceinline__ void st4(float* p, F4 v) { *reinterpret_cast<F4*>(p) = v; }

constexpr float kLnEps = 1e-5f;

// shader-clock timestamp pinned in program order (phase tracing of a kernel; measurement only)
__device__ __forceinline__ unsigned long long clock_pinned() {
#if defined(MLDHIP_SIM)
  return 0ull;
#else
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
#endif
}
__device__ __forceinline__ unsigned long long realtime_100mhz() {
#if defined(MLDHIP_SIM)
  return 0ull;
#else
  return __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace mld
