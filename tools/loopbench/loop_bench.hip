// Stand-alone timing harness for the sample-major persistent loop (kernels/loop_fused.hpp): the kernel alone, on a random weight stream
// of the right shape (9 layers, 50 steps, 1 856 items per step), without the engine around it.  Built in seconds, so a kernel variant is
// one hipcc + one gpurun call:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I motion-latent-diffusion_amd/csrc/kernels [-DLB_VARIANT='true, 0'] [-D...] \
//         -o gpurun_out/lb_x tools/loopbench/loop_bench.hip
//   gpurun -- 'gpurun_out/lb_x [motions=2048] [reps=5] [trace=0|1]'
// Prints one JSON line: ms per launch (min / median), and with trace=1 the kernel's own phase counters (the DBG 5 build of the same
// variant flags: LB_TRACE_VARIANT).  Timing only -- results are checked through the library (tests/, simulator + GPU), never here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "loop_fused.hpp"

#ifndef LB_VARIANT
#define LB_VARIANT true, 0
#endif
#ifndef LB_TRACE_VARIANT
#define LB_TRACE_VARIANT true, 5
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using namespace mld;

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 5, trace = argc > 3 ? atoi(argv[3]) : 0;
  const int L = 9, nb = 4, n = 50, ips = L * kLoopItemsLayer + nb * kLoopItemsSkip, nit = ips + 8;
  const int grid = (B + 7) / 8;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto dev = [&](size_t nfl, float scale, float bias, float** out) {
    std::vector<float> h(nfl);
    for (auto& v : h) v = bias + scale * u(rng);
    if (hipMalloc((void**)out, nfl * sizeof(float)) != hipSuccess) return 1;
    return hipMemcpy(*out, h.data(), nfl * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  float *arena, *stream, *small, *T1, *TP, *init, *lat, *skip, *ddim;
  if (dev((size_t)ips * kLoopItemFloats, 0.06f, 0.f, &arena)) return 1;
  CK(hipMalloc((void**)&stream, (size_t)nit * kLoopItemFloats * sizeof(float)));
  {
    std::vector<LoopItem> items(nit);
    for (int i = 0; i < nit; ++i) items[i] = LoopItem{(long long)(i % ips) * kLoopItemFloats, 32, 0};
    LoopItem* idev;
    CK(hipMalloc((void**)&idev, nit * sizeof(LoopItem)));
    CK(hipMemcpy(idev, items.data(), nit * sizeof(LoopItem), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_loop_stream_kernel<true>, dim3(nit), dim3(512), 0, 0, (const float*)arena, (const LoopItem*)idev, stream);
    CK(hipDeviceSynchronize());
  }
  const size_t small_floats = (size_t)L * kLsLayer + (size_t)nb * 256 + 768;
  {
    std::vector<float> h(small_floats);
    for (auto& v : h) v = 0.02f * u(rng);
    for (int l = 0; l < L; ++l)
      for (int i = 0; i < 256; ++i) h[(size_t)l * kLsLayer + kLsN1W + i] = h[(size_t)l * kLsLayer + kLsN2W + i] = 1.0f + 0.1f * u(rng);
    for (int i = 0; i < 256; ++i) h[(size_t)L * kLsLayer + nb * 256 + i] = 1.0f;
    CK(hipMalloc((void**)&small, small_floats * sizeof(float)));
    CK(hipMemcpy(small, h.data(), small_floats * sizeof(float), hipMemcpyHostToDevice));
  }
  if (dev((size_t)n * 256, 1.f, 0.f, &T1) || dev((size_t)2 * B * 256, 1.f, 0.f, &TP) || dev((size_t)B * 256, 1.f, 0.f, &init)) return 1;
  CK(hipMalloc((void**)&lat, (size_t)B * 256 * sizeof(float)));
  CK(hipMalloc((void**)&skip, (size_t)grid * nb * 48 * 256 * sizeof(float)));
  {
    std::vector<float> h((size_t)n * 4);
    for (int s = 0; s < n; ++s) { h[4 * s] = 0.9f; h[4 * s + 1] = 0.43f; h[4 * s + 2] = 0.92f; h[4 * s + 3] = 0.39f; }
    CK(hipMalloc((void**)&ddim, h.size() * sizeof(float)));
    CK(hipMemcpy(ddim, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  unsigned long long* tr;
  CK(hipMalloc((void**)&tr, 64 * 8 * 16 * sizeof(unsigned long long)));
  CK(hipMemset(tr, 0, 64 * 8 * 16 * sizeof(unsigned long long)));
  LoopArgs a;
  a.stream = stream; a.ips = ips; a.small = small; a.T1 = T1; a.TP = TP; a.init_lat = init; a.lat = lat; a.skip = skip; a.ddim = ddim;
  a.B = B; a.L = L; a.n = n; a.guidance = 7.5f; a.init_sigma = 1.0f; a.trace = tr;
  CK(hipFuncSetAttribute((const void*)den_loop_kernel<LB_VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes));
  CK(hipFuncSetAttribute((const void*)den_loop_kernel<LB_TRACE_VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLoopLdsBytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((den_loop_kernel<LB_VARIANT>), dim3(grid), dim3(512), kLoopLdsBytes, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (r) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  std::vector<float> hl(256);
  CK(hipMemcpy(hl.data(), lat, 256 * sizeof(float), hipMemcpyDeviceToHost));
  double cs = 0; for (float v : hl) cs += v;
  printf("{\"variant\": \"%s\", \"motions\": %d, \"ms_min\": %.3f, \"ms_med\": %.3f, \"lat_checksum\": %.6g", LB_NAME, B, ms.front(), ms[ms.size() / 2], cs);
  if (trace) {
    hipLaunchKernelGGL((den_loop_kernel<LB_TRACE_VARIANT>), dim3(grid), dim3(512), kLoopLdsBytes, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((den_loop_kernel<LB_TRACE_VARIANT>), dim3(grid), dim3(512), kLoopLdsBytes, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    const int nw = std::min(grid, 64);
    std::vector<unsigned long long> h(64 * 8 * 16);
    CK(hipMemcpy(h.data(), tr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double tot = 0, ph[15] = {0};
    for (int w = 0; w < nw * 8; ++w) { tot += (double)h[w * 16 + 15]; for (int k = 0; k < 15; ++k) ph[k] += (double)h[w * 16 + k]; }
    printf(", \"traced_ms\": %.3f, \"cycles_per_wave\": %.0f, \"phase_ms\": [", t, tot / (nw * 8));
    for (int k = 0; k < 12; ++k) printf("%s%.2f", k ? ", " : "", ph[k] / tot * t);
    printf("]");
  }
  printf("}\n");
  return 0;
}
