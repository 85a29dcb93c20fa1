// Stand-alone timing harness for the row-strip GEMMs of the decoder (kernels/gemm_strip_x3.hpp): the in-projection form
// strip_gemm_x3_kernel<6, 1, false, true, true> (N = 768) or, with -DSB_SKIP, the skip-linear form <4, 2, false, false> (K = 512, N = 256).
//   LB_SRC=strip_bench.hip tools/loopbench/build.sh NAME [flags];   build/lb/NAME [motions=1280] [reps=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "gemm_strip_x3.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace mld;
#ifdef SB_SKIP
#ifndef SB_SKIP_RT
#define SB_SKIP_RT 4
#endif
#define SB_VARIANT SB_SKIP_RT, 2, false, false
constexpr int kRT = SB_SKIP_RT, kNSEG = 2, kN = 256; constexpr bool kStage = false;
#else
#ifndef SB_VARIANT
#define SB_VARIANT 6, 1, false, true, true
#endif
#ifndef SB_RT
#define SB_RT 6
#endif
constexpr int kRT = SB_RT, kNSEG = 1, kN = 768; constexpr bool kStage = true;
#endif

int main(int argc, char** argv) {
  const int motions = argc > 1 ? atoi(argv[1]) : 1280, reps = argc > 2 ? atoi(argv[2]) : 5, T = 196, M = motions * T;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto dev = [&](size_t nfl, float scale, float** out) {
    std::vector<float> h(nfl);
    for (auto& v : h) v = scale * u(rng);
    if (hipMalloc((void**)out, nfl * sizeof(float)) != hipSuccess) return 1;
    return hipMemcpy(*out, h.data(), nfl * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  const int nit = (kN / 256) * kNSEG * 16;
  float *arena, *stream, *A, *A2, *Y, *bias;
  if (dev((size_t)nit * kLoopItemFloats, 0.06f, &arena)) return 1;
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
  if (dev((size_t)M * 256, 1.f, &A) || dev((size_t)M * 256, 1.f, &A2) || dev(kN, 0.05f, &bias)) return 1;
  CK(hipMalloc((void**)&Y, (size_t)M * kN * sizeof(float)));
  StripGemmArgs a;
  a.A = A; a.A2 = A2; a.W = stream; a.bias = bias; a.Y = Y; a.ldy = kN; a.M = M; a.N = kN;
  constexpr int lds = strip_gemm_lds_bytes<kRT, kNSEG, kStage>();
  CK(hipFuncSetAttribute((const void*)strip_gemm_x3_kernel<SB_VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((strip_gemm_x3_kernel<SB_VARIANT>), dim3((M + 16 * kRT - 1) / (16 * kRT)), dim3(512), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (r > 1) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  std::vector<float> hy(kN);
  CK(hipMemcpy(hy.data(), Y + (size_t)(M - 1) * kN, kN * sizeof(float), hipMemcpyDeviceToHost));
  double cs = 0; for (float v : hy) cs += v;
  const double gb = (double)M * (256.0 * kNSEG + kN) * 4 / 1e9, flop = 2.0 * M * 256.0 * kNSEG * kN;
  printf("{\"variant\": \"%s\", \"motions\": %d, \"us_min\": %.1f, \"us_med\": %.1f, \"tb_s\": %.2f, \"tflops\": %.1f, \"checksum\": %.6g}\n", LB_NAME, motions, ms.front() * 1e3,
         ms[ms.size() / 2] * 1e3, gb / ms.front(), flop / (ms.front() * 1e-3) / 1e12, cs);
  return 0;
}
