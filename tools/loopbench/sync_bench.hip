// Stand-alone price list for the hand-offs a cluster form of the reverse loop would need ("start with the barrier
// alone").  A GROUP of G workgroups -- all on one XCD (block b runs on XCD b % 8; checked with HW_REG_XCC_ID) or spread over XCDs --
// repeats NPH phases of
//     write my slice (S bytes) -> drain -> arrive on the group's counter -> [weight prefetch in flight] -> poll -> gather all G slices
//     (G x S bytes, L1-bypassing loads) into LDS -> NM matrix instructions per wave
// which is the shape of one column-split GEMM phase whose consumers need full rows.  Two publish forms:
//   PROTO 0  plain stores + vmcnt(0) + relaxed agent counter; consumer sc1 loads      (valid ONLY when the group shares an L2: same XCD)
//   PROTO 1  sc1 (write-through) stores + vmcnt(0) + counter; consumer sc1 loads       (cdna_hip_programming.md Guideline 16 R1: any placement)
// Every gathered word is checked against what its producer must have written in THIS phase (verify = 1), so a stale L1 / L2 line shows
// as a count, not as a timing artefact.  Every spin is bounded (50 ms) and sets an abort word.
// Output: one JSON line per configuration: us per phase (host events), and thread 0's own split into wait / gather / compute (100 MHz clock).
//   build:  LB_SRC=sync_bench.hip tools/loopbench/build.sh sync_bench     run:  build/lb/sync_bench [nphase=2000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SbArgs {
  unsigned* ctr;               // [ngroups] arrival counters, zeroed before every launch
  unsigned* abort_flag;        // [1]
  unsigned* xcc;               // [grid] census: XCC id per workgroup
  unsigned* buf;               // [ngroups][2][G][slice_words]
  const unsigned* wstream;     // weight-stream stand-in
  unsigned long long* stamps;  // [grid][4] 100 MHz ticks summed over phases: wait, gather, compute, produce
  unsigned* errs;              // [2] mismatching words, timeouts
  int G, ngroups, slice_words, nphase, nmfma, wwords, xcd_local, verify, wstream_words;
  int arrive;                  // 0: one counter per group (64 words = 256 B apart), relaxed agent fetch_add; 1: one flag word per producer (the group's flags share a 128-B line), relaxed agent store
  int prefetch_late;           // 0: the weight loads are issued in front of the poll (in flight across the wait), 1: behind it (beside the gather)
};

__device__ __forceinline__ unsigned mk(unsigned phase, unsigned rank, unsigned idx) { return (phase * 2654435761u) ^ (rank << 22) ^ (idx * 40503u + 17u); }

constexpr int kLdsWords = 24 * 1024;   // 96 KB: one workgroup per CU, and room for a 64 KB gather

template <int PROTO>
__global__ __launch_bounds__(512, 2) void sync_bench_kernel(SbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  int group, rank;
  if (a.xcd_local) { const int x = b & 7, i = b >> 3; group = x + 8 * (i / a.G); rank = i % a.G; }
  else { group = b / a.G; rank = b % a.G; }
  if (tid == 0) a.xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // HW_REG_XCC_ID, bits 3:0
  unsigned* ctr = a.ctr + group * 64;
  const size_t gwords = (size_t)2 * a.G * a.slice_words;
  unsigned* gbuf = a.buf + (size_t)group * gwords;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(gbuf, 0, (int)(gwords * 4), 0x00020000);
  unsigned long long t_wait = 0, t_gather = 0, t_comp = 0, t_prod = 0;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  unsigned local_err = 0, sink = 0;
  const int chunks_slice = a.slice_words / 4, chunks_all = chunks_slice * a.G;
  bool dead = false;
  for (int p = 0; p < a.nphase && !dead; ++p) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // ---- produce
    const int half = p & 1;
    const int my_off = (half * a.G + rank) * a.slice_words;     // words
    for (int c = tid; c < chunks_slice; c += 512) {
      const unsigned i0 = c * 4;
      u32x4 v = {mk(p, rank, i0), mk(p, rank, i0 + 1), mk(p, rank, i0 + 2), mk(p, rank, i0 + 3)};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (my_off + i0) * 4, 0, PROTO == 1 ? 16 : 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (a.arrive == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(ctr + rank, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // ---- weight prefetch stand-in: in flight across the wait, or issued behind it
    u32x4 wr[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto prefetch = [&]() __attribute__((always_inline)) {
      const unsigned base = (unsigned)(((size_t)(p * gridDim.x + b) * a.wwords) % (size_t)(a.wstream_words - a.wwords));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = (j * 512 + tid) * 4;
        if (w < a.wwords) wr[j] = *reinterpret_cast<const u32x4*>(a.wstream + (base & ~3u) + w);
      }
    };
    if (!a.prefetch_late) prefetch();
    // ---- wait for the group
    if (tid < 64) {
      const unsigned target = a.arrive == 0 ? (unsigned)a.G * (unsigned)(p + 1) : (unsigned)(p + 1);
      const unsigned* w = a.arrive == 0 ? ctr : ctr + (tid < a.G ? tid : 0);
      const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
      unsigned it = 0;
      bool ok = true;
      for (;;) {
        const bool ready = (a.arrive == 0 && tid != 0) || __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 63u) == 0) {
          unsigned ab = 0;
          if (tid == 0) {
            ab = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab == 0 && __builtin_amdgcn_s_memrealtime() - ts > 5000000ull) {     // 50 ms
              __hip_atomic_store(a.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              atomicAdd(a.errs + 1, 1u);
              ab = 1;
            }
          }
          ab = __builtin_amdgcn_readfirstlane(ab);
          if (ab) { ok = false; break; }
        }
      }
      if (tid == 0) smem[kLdsWords - 1] = ok ? 1u : 0u;
    }
    __syncthreads();
    if (smem[kLdsWords - 1] == 0u) { dead = true; break; }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (a.prefetch_late) prefetch();
    // ---- gather every slice of the group (L1-bypassing loads), 8 loads in flight per lane, into LDS
    const int goff = half * a.G * a.slice_words;
    for (int c0 = tid; c0 < chunks_all; c0 += 512 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * 512;
        v[j] = c < chunks_all ? __builtin_amdgcn_raw_buffer_load_b128(rs, (goff + c * 4) * 4, 0, 16) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * 512;
        if (c < chunks_all) {
          *reinterpret_cast<u32x4*>(smem + ((c * 4) % (kLdsWords - 64))) = v[j];
          if (a.verify) {
            const unsigned r = c / chunks_slice, i0 = (c % chunks_slice) * 4;
            local_err += (v[j][0] != mk(p, r, i0)) + (v[j][1] != mk(p, r, i0 + 1)) + (v[j][2] != mk(p, r, i0 + 2)) + (v[j][3] != mk(p, r, i0 + 3));
          } else {
            sink ^= v[j][0] ^ v[j][3];
          }
        }
      }
    }
    __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    // ---- compute stand-in: NM dependent-by-four matrix instructions per wave on what was gathered / prefetched
    {
      const u32x4 x0 = *reinterpret_cast<const u32x4*>(smem + ((tid * 4) % (kLdsWords - 64)));
      u32x4 wx = wr[0] ^ wr[1] ^ wr[2] ^ wr[3];
      wx = (wx & 0x03ff03ffu) | 0x3c003c00u;      // halves in [1, 2)
      const u32x4 xx = (x0 & 0x03ff03ffu) | 0x38003800u;
      for (int i = 0; i < a.nmfma; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wx), __builtin_bit_cast(f16x8, xx), acc[j], 0, 0, 0);
      }
      asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    t_prod += t1 - t0; t_wait += t2 - t1; t_gather += t3 - t2; t_comp += t4 - t3;
  }
  if (local_err) atomicAdd(a.errs, local_err);
  if (tid == 0) {
    unsigned long long* o = a.stamps + (size_t)b * 4;
    o[0] = t_wait; o[1] = t_gather; o[2] = t_comp; o[3] = t_prod;
  }
  if (sink == 0x12345u || acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) a.errs[0] += 1u << 30;   // keep the results alive
}

// the launch-per-phase baseline on the same box: the same gather + compute + slice store as ONE kernel of `grid` workgroups, NPH dependent launches
__global__ __launch_bounds__(512, 2) void phase_launch_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, const unsigned* __restrict__ wstream,
                                                              int read_words, int slice_words, int nmfma, int wwords, int woff) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  u32x4 wr[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = (j * 512 + tid) * 4;
    if (w < wwords) wr[j] = *reinterpret_cast<const u32x4*>(wstream + (size_t)woff + (size_t)b * wwords + w);
  }
  const int chunks = read_words / 4;
  for (int c0 = tid; c0 < chunks; c0 += 512 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = c0 + j * 512; v[j] = c < chunks ? *reinterpret_cast<const u32x4*>(in + c * 4) : u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = c0 + j * 512; if (c < chunks) *reinterpret_cast<u32x4*>(smem + ((c * 4) % (kLdsWords - 64))) = v[j]; }
  }
  __syncthreads();
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const u32x4 x0 = *reinterpret_cast<const u32x4*>(smem + ((tid * 4) % (kLdsWords - 64)));
  u32x4 wx = wr[0] ^ wr[1] ^ wr[2] ^ wr[3];
  wx = (wx & 0x03ff03ffu) | 0x3c003c00u;
  const u32x4 xx = (x0 & 0x03ff03ffu) | 0x38003800u;
  for (int i = 0; i < nmfma; i += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wx), __builtin_bit_cast(f16x8, xx), acc[j], 0, 0, 0);
  }
  const unsigned r = __builtin_bit_cast(unsigned, acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
  for (int c = tid; c < slice_words / 4; c += 512) {
    u32x4 v = {r, r + 1, r + 2, r + 3};
    *reinterpret_cast<u32x4*>(out + (size_t)b * slice_words + c * 4) = v;
  }
}

struct Cfg { int G, slice_bytes, ngroups, nmfma, wbytes; const char* what; };

int main(int argc, char** argv) {
  const int nphase = argc > 1 ? atoi(argv[1]) : 2000;
  const int wstream_words = 8 << 20;   // 32 MB
  unsigned *ctr, *abortf, *xcc, *buf, *wstream, *errs;
  unsigned long long* stamps;
  CK(hipMalloc((void**)&ctr, 256 * 64 * 4));
  CK(hipMalloc((void**)&abortf, 64));
  CK(hipMalloc((void**)&xcc, 256 * 4));
  CK(hipMalloc((void**)&errs, 64));
  CK(hipMalloc((void**)&stamps, 256 * 4 * 8));
  const size_t buf_bytes = (size_t)64 << 20;
  CK(hipMalloc((void**)&buf, buf_bytes));
  CK(hipMalloc((void**)&wstream, (size_t)wstream_words * 4));
  {
    std::vector<unsigned> h(wstream_words);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    CK(hipMemcpy(wstream, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute((const void*)sync_bench_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWords * 4));
  CK(hipFuncSetAttribute((const void*)sync_bench_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWords * 4));
  CK(hipFuncSetAttribute((const void*)phase_launch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWords * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const Cfg cfgs[] = {
      {4, 16, 8, 0, 0, "sync only, 4 per group"},
      {12, 16, 8, 0, 0, "sync only, 12 per group"},
      {32, 16, 8, 0, 0, "sync only, 32 per group (a whole XCD)"},
      {4, 16, 24, 0, 0, "sync only, 24 groups of 4"},
      {4, 4096, 24, 64, 32768, "token x column form: attention-output / FFN2-output gather (16 KB from 4)"},
      {4, 16384, 24, 64, 32768, "token x column form: hidden gather (64 KB from 4)"},
      {12, 4096, 8, 64, 32768, "token x column form: layer-output gather (48 KB from 12)"},
      {4, 49152, 8, 256, 32768, "head-split form: four partial slabs (192 KB from 4)"},
      {8, 6144, 8, 128, 32768, "8 per group: 48 KB"},
      {8, 24576, 8, 128, 32768, "8 per group: 192 KB"},
      {16, 3072, 8, 64, 32768, "16 per group: 48 KB"},
      {16, 12288, 8, 64, 32768, "16 per group: 192 KB"},
      {32, 1536, 8, 32, 32768, "XCD-wide column split: 48 KB"},
      {32, 6144, 8, 32, 32768, "XCD-wide column split: 192 KB"},
  };
  printf("{\"nphase\": %d, \"rows\": [\n", nphase);
  bool first = true;
  for (const Cfg& c : cfgs) {
    for (int xl = 1; xl >= 0; --xl)
      for (int proto = 0; proto < 2; ++proto)
        for (int variant = 0; variant < 4; ++variant) {
          const int verify = variant == 0, arrive = variant >= 2, late = variant == 3;
          if (c.nmfma == 0 && (proto == 1 || late)) continue;
          SbArgs a;
          a.ctr = ctr; a.abort_flag = abortf; a.xcc = xcc; a.buf = buf; a.wstream = wstream; a.stamps = stamps; a.errs = errs;
          a.G = c.G; a.ngroups = c.ngroups; a.slice_words = c.slice_bytes / 4; a.nphase = nphase; a.nmfma = c.nmfma; a.wwords = c.wbytes / 4;
          a.xcd_local = xl; a.verify = verify; a.wstream_words = wstream_words; a.arrive = arrive; a.prefetch_late = late;
          const int grid = c.G * c.ngroups;
          if (grid > 256 || (size_t)c.ngroups * 2 * c.G * c.slice_bytes > buf_bytes) continue;
          CK(hipMemset(ctr, 0, 256 * 64 * 4));
          CK(hipMemset(abortf, 0, 64));
          CK(hipMemset(errs, 0, 64));
          CK(hipMemset(xcc, 0xff, 256 * 4));
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, 0));
          if (proto == 0) hipLaunchKernelGGL(sync_bench_kernel<0>, dim3(grid), dim3(512), kLdsWords * 4, 0, a);
          else hipLaunchKernelGGL(sync_bench_kernel<1>, dim3(grid), dim3(512), kLdsWords * 4, 0, a);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          CK(hipGetLastError());
          float ms = 0.f;
          CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<unsigned long long> st(256 * 4);
          std::vector<unsigned> hx(256), he(16);
          CK(hipMemcpy(st.data(), stamps, 256 * 4 * 8, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(he.data(), errs, 64, hipMemcpyDeviceToHost));
          double sw = 0, sg = 0, sc = 0, sp = 0;
          for (int b = 0; b < grid; ++b) { sw += st[b * 4]; sg += st[b * 4 + 1]; sc += st[b * 4 + 2]; sp += st[b * 4 + 3]; }
          const double k = 0.01 / ((double)grid * nphase);     // 100 MHz ticks -> us per phase
          int mixed = 0;                                         // groups whose members sit on more than one XCD
          for (int g = 0; g < c.ngroups; ++g) {
            int x0 = -1;
            bool mix = false;
            for (int b = 0; b < grid; ++b) {
              const int gg = xl ? (b & 7) + 8 * ((b >> 3) / c.G) : b / c.G;
              if (gg != g) continue;
              if (x0 < 0) x0 = (int)hx[b];
              else if ((int)hx[b] != x0) mix = true;
            }
            mixed += mix;
          }
          printf("%s {\"what\": \"%s\", \"G\": %d, \"slice_bytes\": %d, \"gather_bytes\": %d, \"groups\": %d, \"workgroups\": %d, \"same_xcd_map\": %d, \"groups_spanning_xcds\": %d, "
                 "\"proto\": \"%s\", \"arrive\": \"%s\", \"prefetch\": \"%s\", \"verify\": %d, \"nmfma\": %d, \"prefetch_bytes\": %d, \"us_per_phase\": %.3f, \"t0_us\": {\"produce\": %.3f, \"wait\": %.3f, \"gather\": %.3f, \"compute\": %.3f}, "
                 "\"stale_words\": %u, \"timeouts\": %u}",
                 first ? "" : ",\n", c.what, c.G, c.slice_bytes, c.G * c.slice_bytes, c.ngroups, grid, xl, mixed, proto ? "sc1 stores + sc1 loads" : "plain stores + sc1 loads", arrive ? "flag per producer" : "counter", late ? "behind the poll" : "in front of the poll",
                 verify, c.nmfma, c.wbytes, ms * 1000.0 / nphase, sp * k, sw * k, sg * k, sc * k, he[0] & 0x3fffffffu, he[1]);
          first = false;
          fflush(stdout);
        }
  }
  // ---- launches: the same body as NPH dependent launches (ping-pong buffers)
  for (const Cfg& c : cfgs) {
    if (c.nmfma == 0) continue;
    const int grid = c.G * c.ngroups;
    for (int i = 0; i < 20; ++i)
      hipLaunchKernelGGL(phase_launch_kernel, dim3(grid), dim3(512), kLdsWords * 4, 0, (const unsigned*)buf, buf + (16 << 20) / 4, (const unsigned*)wstream, c.G * c.slice_bytes / 4,
                         c.slice_bytes / 4, c.nmfma, c.wbytes / 4, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int p = 0; p < nphase; ++p) {
      unsigned* in = buf + ((p & 1) ? (16 << 20) / 4 : 0);
      unsigned* out = buf + ((p & 1) ? 0 : (16 << 20) / 4);
      hipLaunchKernelGGL(phase_launch_kernel, dim3(grid), dim3(512), kLdsWords * 4, 0, (const unsigned*)in, out, (const unsigned*)wstream, c.G * c.slice_bytes / 4, c.slice_bytes / 4,
                         c.nmfma, c.wbytes / 4, (int)(((size_t)p * grid * (c.wbytes / 4)) % (size_t)(wstream_words - 256 * 8192)));
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(",\n {\"what\": \"LAUNCH per phase: %s\", \"workgroups\": %d, \"gather_bytes\": %d, \"nmfma\": %d, \"us_per_phase\": %.3f}", c.what, grid, c.G * c.slice_bytes, c.nmfma,
           ms * 1000.0 / nphase);
    fflush(stdout);
  }
  printf("\n]}\n");
  return 0;
}
