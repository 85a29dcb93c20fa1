// Stand-alone timing harness for the one-launch decoder tail (kernels/ffn_strip.hpp, ffn_strip_x3_kernel<3, true, true>): the kernel alone on
// random operands of the right shape (M = motions x 196 frame rows), like loop_bench.hip for the persistent loop.  Timing only.
//   tools/loopbench/build.sh NAME [flags]  with  LB_SRC=tail_bench.hip;   build/lb/NAME [motions=1280] [reps=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "ffn_strip.hpp"

#ifndef TB_VARIANT
#define TB_VARIANT 3, true, true
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace mld;

int main(int argc, char** argv) {
  const int motions = argc > 1 ? atoi(argv[1]) : 1280, reps = argc > 2 ? atoi(argv[2]) : 5, T = 196;
  const int M = motions * T;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto dev = [&](size_t nfl, float scale, float bias, float** out) {
    std::vector<float> h(nfl);
    for (auto& v : h) v = bias + scale * u(rng);
    if (hipMalloc((void**)out, nfl * sizeof(float)) != hipSuccess) return 1;
    return hipMemcpy(*out, h.data(), nfl * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  const int nit = kFfnStripItems + 16;
  float *arena, *stream, *AO, *res, *Y, *b1, *b2, *gam, *bet, *bo, *g1, *be1, *g2, *be2, *cvec;
  if (dev((size_t)nit * kLoopItemFloats, 0.06f, 0.f, &arena)) return 1;
  CK(hipMalloc((void**)&stream, (size_t)nit * kLoopItemFloats * sizeof(float)));
  {
    std::vector<LoopItem> items(nit);
    for (int i = 0; i < nit; ++i) items[i] = LoopItem{(long long)i * kLoopItemFloats, 32, 0};
    LoopItem* idev;
    CK(hipMalloc((void**)&idev, nit * sizeof(LoopItem)));
    CK(hipMemcpy(idev, items.data(), nit * sizeof(LoopItem), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_loop_stream_kernel<true>, dim3(nit), dim3(512), 0, 0, (const float*)arena, (const LoopItem*)idev, stream);
    CK(hipDeviceSynchronize());
  }
  if (dev((size_t)M * 256, 1.f, 0.f, &AO) || dev((size_t)M * 256, 1.f, 0.f, &res)) return 1;
  CK(hipMalloc((void**)&Y, (size_t)M * 256 * sizeof(float)));
  if (dev(1024, 0.05f, 0.f, &b1) || dev(256, 0.05f, 0.f, &b2) || dev(256, 0.1f, 1.f, &gam) || dev(256, 0.05f, 0.f, &bet) || dev(256, 0.05f, 0.f, &bo) ||
      dev(256, 0.1f, 1.f, &g1) || dev(256, 0.05f, 0.f, &be1) || dev(256, 0.1f, 1.f, &g2) || dev(256, 0.05f, 0.f, &be2) || dev((size_t)motions * 256, 1.f, 0.f, &cvec)) return 1;
  FfnArgs a;
  a.W1 = stream + (size_t)16 * kLoopItemFloats; a.b1 = b1; a.b2 = b2; a.gamma = gam; a.beta = bet; a.Y = Y; a.M = M;
  a.AO = AO; a.Wo = stream; a.bo = bo; a.res = res; a.g1 = g1; a.be1 = be1; a.cvec = cvec; a.rpg = T; a.g2 = g2; a.be2 = be2;
  constexpr int lds = ffn_strip_lds_bytes<3>();
  CK(hipFuncSetAttribute((const void*)ffn_strip_x3_kernel<TB_VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ffn_strip_x3_kernel<TB_VARIANT>), dim3((M + 47) / 48), dim3(512), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (r > 1) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  std::vector<float> hy(256);
  CK(hipMemcpy(hy.data(), Y + (size_t)(M - 1) * 256, 256 * sizeof(float), hipMemcpyDeviceToHost));
  double cs = 0; for (float v : hy) cs += v;
  const double flop = 2.0 * M * (256.0 * 256 + 2.0 * 256 * 1024);
  printf("{\"variant\": \"%s\", \"motions\": %d, \"us_min\": %.1f, \"us_med\": %.1f, \"tflops\": %.1f, \"checksum\": %.6g}\n", LB_NAME, motions, ms.front() * 1e3, ms[ms.size() / 2] * 1e3,
         flop / (ms.front() * 1e-3) / 1e12, cs);
  return 0;
}
