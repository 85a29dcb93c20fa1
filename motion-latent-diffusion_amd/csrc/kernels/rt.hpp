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
// every workgroup of the grid resident at once (kernels whose workgroups hand data to each other: loop_cluster.hpp); on the GPU a plain
// launch whose grid the caller sized to the chip
#define MLD_LAUNCH_CORESIDENT(kernel, grid, block, shmem, stream, ...) \
  hipsim::launch_coresident((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define MLD_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define MLD_LAUNCH_CORESIDENT MLD_LAUNCH
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

// a value that is the same in every lane of the wave (e.g. threadIdx.x >> 6), moved to an SGPR so that branches on it are scalar
__device__ __forceinline__ int wave_uniform(int v) {
#if defined(MLDHIP_SIM)
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
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

#if !defined(MLDHIP_SIM)
// DPP lane permutes (VALU speed, no LDS crossbar): quad_perm / row_half_mirror / row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#endif

// sum over the 16 lanes that share (lane >> 4): the column axis of an MFMA C tile (= one DPP "row").
// Every lane of the row receives the total.
__device__ __forceinline__ float sum16(float v) {
#if defined(MLDHIP_SIM)
  v += wave_xor(v, 1);
  v += wave_xor(v, 2);
  v += wave_xor(v, 4);
  v += wave_xor(v, 8);
#else
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror: lane i <-> 7-i inside each 8
  v += dpp_mov<0x140>(v);   // row_mirror:      lane i <-> 15-i
#endif
  return v;
}
// v + (v of lane ^ 16) and v + (v of lane ^ 32) on the gfx950 lane-swap instructions (v_permlane16_swap / v_permlane32_swap: odd 16-lane
// rows of one register <-> even rows of the other; upper 32 lanes <-> lower 32 lanes): with both operands = v the two results are
// {v of the even row / lower half, v of the odd row / upper half} in every lane, so their sum is the xor-16 / xor-32 butterfly -- VALU only.
// (__shfl_xor across 16-lane rows is ds_bpermute_b32: an LDS-crossbar round trip per shuffle on the dependent chain of every LayerNorm
// statistic and attention score of the persistent loop: r04 loop 20.55 -> 19.70 ms at 1 280 motions, attention phase 1.82 -> 1.37 ms.)
// Inline assembly, not __builtin_amdgcn_permlane16_swap: hipcc (ROCm 7.2) reads BOTH results of the builtin from its first result
// register when they feed one v_add_f32 (build/t/swap_test2: 576 of 576 sums wrong); the s_nop covers the VALU-write -> lane-swap-read
// hazard the compiler would otherwise pad itself (it does not look inside inline assembly).
__device__ __forceinline__ float add_xor16(float v) {
#if defined(MLDHIP_SIM)
  return v + wave_xor(v, 16);
#else
  float a = v, c = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));
  return a + c;
#endif
}
__device__ __forceinline__ float add_xor32(float v) {
#if defined(MLDHIP_SIM)
  return v + wave_xor(v, 32);
#else
  float a = v, c = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
  return a + c;
#endif
}
// sum / max over the 4 lanes {l, l^16, l^32, l^48}: the row-group axis of an MFMA C tile
#ifndef MLD_SUM_GROUPS_BPERMUTE
#define MLD_SUM_GROUPS_BPERMUTE 0
#endif
__device__ __forceinline__ float sum_groups(float v) {
#if MLD_SUM_GROUPS_BPERMUTE
  v += wave_xor(v, 16);
  v += wave_xor(v, 32);
  return v;
#else
  return add_xor32(add_xor16(v));
#endif
}
__device__ __forceinline__ float max_groups(float v) {
  v = fmaxf(v, wave_xor(v, 16));
  v = fmaxf(v, wave_xor(v, 32));
  return v;
}
// sum over all 64 lanes, result in every lane: 4 DPP adds inside each 16-lane row, then the four row
// totals are read through SGPRs (v_readlane) -- no ds_bpermute on the dependent chain.
__device__ __forceinline__ float sum64(float v) {
  v = sum16(v);
#if defined(MLDHIP_SIM) || defined(MLD_SUM64_SWAP)
  return sum_groups(v);
#else
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
#endif
}

// max over all 64 lanes, result in every lane (same DPP + readlane scheme as sum64)
__device__ __forceinline__ float max64(float v) {
#if defined(MLDHIP_SIM)
  v = fmaxf(v, wave_xor(v, 1)); v = fmaxf(v, wave_xor(v, 2)); v = fmaxf(v, wave_xor(v, 4)); v = fmaxf(v, wave_xor(v, 8));
  return max_groups(v);
#else
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
#endif
}

// 1 / x and e^x on the hardware's 1-ulp transcendental units (v_rcp_f32, v_exp_f32): libm's expf and IEEE division expand to
// 6-10 instructions each (range handling, Newton step, div_fixup), which is what the epilogues of the sample-major loop and of the
// decoder's feed-forward kernels were spending their VALU time on (r03: the split-mode loop ran 27.9 of its 35 ms WITHOUT any weight
// traffic).  Arguments here are finite and the results feed fp32 arithmetic with >= 1e-7 of its own rounding.
__device__ __forceinline__ float fast_rcp(float x) {
#if defined(MLDHIP_SIM)
  return 1.0f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
// 2^x (v_exp_f32) and "true in any lane" (wave-uniform)
__device__ __forceinline__ float fast_exp2(float x) {
#if defined(MLDHIP_SIM)
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ bool wave_any(bool p) {
#if defined(MLDHIP_SIM)
  float f = p ? 1.f : 0.f;
  for (int m = 1; m < 64; m <<= 1) f = fmaxf(f, hipsim::shfl_xor(f, m));
  return f > 0.f;
#else
  return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
}
__device__ __forceinline__ float fast_exp(float x) {
#if defined(MLDHIP_SIM)
  return expf(x);
#else
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#endif
}

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26), branch free: the epilogue of the FFN1 GEMMs
// evaluates it 4-32 times per lane and libm's erff is a divergent multi-branch routine.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = fast_rcp(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = 1.0f - p * t * fast_exp(-ax * ax);
  return copysignf(y, x);
}

// GELU(x) = x/2 (1 + erf(x/sqrt2)) on the same erf (A&S 7.1.26, z = |x|/sqrt2), arranged for the epilogues that evaluate it once per
// hidden activation: erf(z) = sign(x) y, y = 1 - p(t) t e^(-z^2), so GELU = h + |h| y with h = x/2 -- no copysign, no (1 + erf),
// and e^(-z^2) = 2^(-w^2), w = x sqrt(log2(e)/2), straight on v_exp_f32: 12 VALU + 2 transcendental instructions (15 + 2 before).
__device__ __forceinline__ float gelu_erf(float x) {
  const float h = 0.5f * x;
  const float t = fast_rcp(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float w = x * 0.84932180028801904272f;           // sqrt(log2(e) / 2)
#if defined(MLDHIP_SIM)
  const float e = exp2f(-(w * w));
#else
  const float e = __builtin_amdgcn_exp2f(-(w * w));
#endif
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = fmaf(-e, p * t, 1.0f);
  return fmaf(fabsf(h), y, h);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

struct alignas(16) F4 { float x, y, z, w; };
__device__ __forceinline__ F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }
__device__ __forceinline__ void st4(float* p, F4 v) { *reinterpret_cast<F4*>(p) = v; }
// streaming ("nt") forms for activations that one workgroup touches once per launch (the decoder's row strips): `nt` on the
// global_load / global_store, i.e. no reuse expected -- the weight streams every workgroup re-reads keep the default policy.
// Option "nt_hints" (state.hpp; a template parameter of the kernels that take it); the simulator has no cache model: plain accesses there.
__device__ __forceinline__ F4 ld4_nt(const float* p) {
#if defined(MLDHIP_SIM)
  return ld4(p);
#else
  return __builtin_bit_cast(F4, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)));
#endif
}
__device__ __forceinline__ void st4_nt(float* p, F4 v) {
#if defined(MLDHIP_SIM)
  st4(p, v);
#else
  __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(p));
#endif
}
template <bool NT_>
__device__ __forceinline__ F4 ld4_hint(const float* p) { if constexpr (NT_) return ld4_nt(p); else return ld4(p); }
template <bool NT_>
__device__ __forceinline__ void st1_hint(float* p, float v) {
#if !defined(MLDHIP_SIM)
  if constexpr (NT_) { __builtin_nontemporal_store(v, p); return; }
#endif
  *p = v;
}
template <bool NT_>
__device__ __forceinline__ void st4_hint(float* p, F4 v) { if constexpr (NT_) st4_nt(p, v); else st4(p, v); }

constexpr float kLnEps = 1e-5f;

// ---- split 16-bit arithmetic ("x3"): x = hi + lo with hi = f16(x), lo = f16(x - hi) keeps 22 mantissa bits;
// x*y ~ lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16 (fp32 accumulate) costs 3 matrix instructions per 32-wide K chunk
// instead of 8 fp32 ones at 1/16 the rate: ~5x the fp32 MFMA throughput at ~5e-7 relative error per product (fp32: 6e-8, plain
// bf16: 4e-3).  Rounds 1-2 split into bf16 halves (16 mantissa bits, 1.5e-5 per product): same instruction count and rate,
// 30x the error -- 50 guided steps amplified that to 1.1e-3 on the joints and ruled the reverse loop out; IEEE half does not
// (oracle-side emulation: latents 1.5e-4 from fp64 against 8e-5 for fp32 and 2.4e-3 for the bf16 split).  What half gives up
// is range: hi saturates at +-65 504 (activations here are post-LayerNorm / post-GELU / attention outputs, |x| < ~100; weights
// < 10) and low parts of |x| < 0.125 fall into f16 subnormals (absolute error <= 3e-8, kept: hipcc's default mode preserves
// f16 denormals on the conversions and on MFMA inputs).
struct alignas(16) U4 { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {      // round-to-nearest-even, result in bits 0..15
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned f16_rne_bits(float x) {       // IEEE half, round-to-nearest-even, in bits 0..15
  const _Float16 h = (_Float16)x;
  return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_bits_value(unsigned b) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(b & 0xFFFFu));
}
// two floats -> packed half pair of the high parts and of the low parts (element 0 in the low half of the word), NO range clamp:
// five instructions on gfx950 (v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32).  For values produced inside
// a kernel (GELU / LayerNorm / attention outputs): anything beyond +-65 504 becomes inf and then NaN, loudly, instead of saturating.
__device__ __forceinline__ void split16_two(float a, float b, unsigned& hi, unsigned& lo) {
#if defined(MLDHIP_SIM)
  const unsigned ha = f16_rne_bits(a), hb = f16_rne_bits(b);
  hi = ha | (hb << 16);
  lo = f16_rne_bits(a - f16_bits_value(ha)) | (f16_rne_bits(b - f16_bits_value(hb)) << 16);
#else
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  const f16x2_t h = __builtin_convertvector(v, f16x2_t);
  const f32x2_t rem = v - __builtin_convertvector(h, f32x2_t);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, f16x2_t));
#endif
}
// split16_two in two pieces (epilogues that spread the work between matrix instructions): the packed high halves, then the packed low ones
__device__ __forceinline__ unsigned split16_hi(float a, float b) {
#if defined(MLDHIP_SIM)
  return f16_rne_bits(a) | (f16_rne_bits(b) << 16);
#else
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
#endif
}
__device__ __forceinline__ unsigned split16_lo(float a, float b, unsigned hi) {
#if defined(MLDHIP_SIM)
  return f16_rne_bits(a - f16_bits_value(hi & 0xFFFFu)) | (f16_rne_bits(b - f16_bits_value(hi >> 16)) << 16);
#else
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  const f32x2_t rem = v - __builtin_convertvector(__builtin_bit_cast(f16x2_t, hi), f32x2_t);
  return __builtin_bit_cast(unsigned, __builtin_convertvector(rem, f16x2_t));
#endif
}
// the same for values that arrive from outside (input rows, weights): clamped to the half range first
__device__ __forceinline__ void split16_pair(float a, float b, unsigned& hi, unsigned& lo) {
  split16_two(fminf(fmaxf(a, -65504.f), 65504.f), fminf(fmaxf(b, -65504.f), 65504.f), hi, lo);
}
// one float -> (high, low) half bits (kernels that write 16-bit elements of a split image one at a time)
__device__ __forceinline__ void split16_one(float a, unsigned short& hi, unsigned short& lo) {
  const float ca = fminf(fmaxf(a, -65504.f), 65504.f);
  const unsigned h = f16_rne_bits(ca);
  hi = (unsigned short)h;
  lo = (unsigned short)f16_rne_bits(ca - f16_bits_value(h));
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[row = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col = l&15],
// j = 0..7 packed little-endian in 4 dwords; C/D as for the fp32 form.
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(U4 a, U4 b, f32x4 c) {
#if defined(MLDHIP_SIM)
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  return hipsim::mfma_bf16_16x16x32(av, bv, c);
#else
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x32_f16: the same operand layout with IEEE half elements (the split arithmetic above)
__device__ __forceinline__ f32x4 mfma_x3_16x16x32(U4 a, U4 b, f32x4 c) {
#if defined(MLDHIP_SIM)
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  return hipsim::mfma_f16_16x16x32(av, bv, c);
#else
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#endif
}

// ---- reduced-precision operand formats of the GEMM kernels (operands only: accumulation, bias, residual, LayerNorm,
// softmax and every stored activation stay fp32).  PREC codes shared by gemm.hpp / tile32.hpp / strip.hpp:
enum : int { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_BF16 = 2 };      // (3 was PREC_FP8: retired in round 6 with the mode, include/mldhip.h)

// two floats -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950), element 0 in the low half
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
#if defined(MLDHIP_SIM)
  return bf16_rne_bits(a) | (bf16_rne_bits(b) << 16);
#else
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
#endif
}
struct alignas(8) U2 { unsigned x, y; };
// ds_read_b64_tr_b16: the LDS transpose read of gfx950.  Inside each 16-lane group the lanes' 8-byte reads form a [4][16] block of
// 16-bit elements (lane L supplies row L / 4, columns 4 (L % 4) .. + 3: its own 8-byte aligned address, any row stride) and lane i
// receives COLUMN i: element j = row j.  With V staged ROW-major [key][dim] this yields an MFMA B fragment (four keys of one head
// dim per lane) without a transposed image (attention.hpp).  Returns the four halves as two words {(e0, e1), (e2, e3)}.
__device__ __forceinline__ U2 lds_read_tr16_b64(const unsigned* p) {
#if defined(MLDHIP_SIM)
  U2 r;
  hipsim::ds_read_tr16_b64(p, &r.x);
  return r;
#else
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  const v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)p);
  return __builtin_bit_cast(U2, v);
#endif
}
// scheduling fence: nothing moves across it.  The register-direct weight rings of loop_fused.hpp / ffn_strip.hpp depend on it -- without
// a fence after every item, hipcc's scheduler sinks each `global_load` of the ring down to its first use (it trades the prefetch
// distance for register pressure) and the kernels run one L2 round trip per item (r03: s_waitcnt vmcnt(0) behind every load).
__device__ __forceinline__ void sched_fence() {
#if !defined(MLDHIP_SIM)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

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
// shader clock without draining outstanding global loads (stamps inside a software-pipelined loop)
__device__ __forceinline__ unsigned long long clock_light() {
#if defined(MLDHIP_SIM)
  return 0ull;
#else
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
#endif
}

// ---- inter-workgroup hand-off inside one launch (kernels/loop_cluster.hpp; cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md
// "Workgroup dispatch, XCD placement & inter-workgroup visibility").  A CU's vector L1 is never refreshed by another CU's stores and the
// eight XCD L2s are not coherent with each other, so: payload stores are WRITE-THROUGH (sc1: valid for any placement) or plain (valid only
// when producer and consumer share an XCD's L2 -- the kernel checks HW_REG_XCC_ID before it uses them), every storing wave drains its
// memory counter, ONE lane publishes a flag with a relaxed agent-scope store, the consumer polls that word relaxed (sc1) and then reads the
// payload with L1-bypassing (sc1) loads.  All shared data goes through ONE buffer descriptor (byte offsets): the cache-policy bits are an
// immediate of the buffer instructions, which hipcc counts in its wait bookkeeping (inline-asm loads are invisible to it).
// The simulator has no caches: plain accesses, and a spinning fiber yields.
#if defined(MLDHIP_SIM)
struct XBuf { unsigned char* base; };
__device__ __forceinline__ XBuf xbuf_make(void* p, unsigned) { return XBuf{static_cast<unsigned char*>(p)}; }
__device__ __forceinline__ F4 xbuf_ld4(const XBuf& b, unsigned off) { return *reinterpret_cast<const F4*>(b.base + off); }
template <bool WT> __device__ __forceinline__ void xbuf_st4(const XBuf& b, unsigned off, F4 v) { *reinterpret_cast<F4*>(b.base + off) = v; }
template <bool WT> __device__ __forceinline__ void xbuf_st2(const XBuf& b, unsigned off, U2 v) { *reinterpret_cast<U2*>(b.base + off) = v; }
__device__ __forceinline__ unsigned flag_load(const unsigned* p) { return *reinterpret_cast<const volatile unsigned*>(p); }
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) { *reinterpret_cast<volatile unsigned*>(p) = v; }
__device__ __forceinline__ void host_flag_store(unsigned* p, unsigned v) { if (p) *reinterpret_cast<volatile unsigned*>(p) = v; }
__device__ __forceinline__ void poll_fence() {}
__device__ __forceinline__ void spin_pause() { hipsim::yield(); }
__device__ __forceinline__ void drain_stores() {}
__device__ __forceinline__ unsigned xcc_id() { return 0u; }
#else
typedef __amdgpu_buffer_rsrc_t XBuf;
__device__ __forceinline__ XBuf xbuf_make(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ F4 xbuf_ld4(const XBuf& b, unsigned off) { return __builtin_bit_cast(F4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 16)); }      // aux 16 = sc1: L2-served
template <bool WT> __device__ __forceinline__ void xbuf_st4(const XBuf& b, unsigned off, F4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), b, (int)off, 0, WT ? 16 : 0); }
template <bool WT> __device__ __forceinline__ void xbuf_st2(const XBuf& b, unsigned off, U2 v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), b, (int)off, 0, WT ? 16 : 0); }
__device__ __forceinline__ unsigned flag_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a word in pinned HOST memory the host reads without a device synchronisation (the cluster loop's sticky "a wait timed out" word): system scope
__device__ __forceinline__ void host_flag_store(unsigned* p, unsigned v) { if (p) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// behind a successful poll, in front of the payload loads: the relaxed flag load does not order later loads for the COMPILER (the hardware issues a wave's loads in
// order and the payloads are read with sc1 loads served by the L2 the flag came from) -- a compiler-only fence keeps hipcc from hoisting them above the poll loop
__device__ __forceinline__ void poll_fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }      // HW_REG_XCC_ID bits 3:0
#endif

__device__ __forceinline__ unsigned long long realtime_100mhz() {
#if defined(MLDHIP_SIM)
  return 0ull;
#else
  return __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace mld
