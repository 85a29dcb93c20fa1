// Stand-alone timing harness for the LDS-staged GEMM family (kernels/gemm.hpp) at the shapes of the diffusion-only variant
// (config 4: M = 2 * 64 * 196 = 25 088 rows of width 512; N in {512, 1024, 1536}, K in {512, 1024}), split-f16 operands.
//   LB_SRC=gemm_bench.hip tools/loopbench/build.sh NAME -DGB_TILE="4, 2, 4, 4" [-DGB_K=512 -DGB_N=512];   build/lb/NAME [M=25088] [reps=5]
// Prints the time, the TFLOP/s (algorithmic, x3 executed) and a checksum of the last output row plus its largest deviation from a
// double-precision host product (so every tile shape is checked, not only timed).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "gemm.hpp"
#include "gemm_pipe.hpp"
#include "elementwise.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace mld;
#ifndef GB_TILE
#define GB_TILE 2, 4, 2, 2
#endif
#ifndef GB_K
#define GB_K 512
#endif
#ifndef GB_N
#define GB_N 512
#endif
#ifndef GB_RD
#define GB_RD 3
#endif
#ifdef GB_PIPE      // the software-pipelined big-tile kernel (kernels/gemm_pipe.hpp), 1-D XCD-aware grid
#define GB_KERNEL gemm_pipe_x3_kernel<GB_TILE, GB_K / 32, GB_RD>
#else
#define GB_KERNEL gemm_kernel<GB_TILE, false, true, PREC_BF16X3, GB_K / 32>
#endif
constexpr int kTile[4] = {GB_TILE};
constexpr int kBM = kTile[0] * kTile[2] * 16, kBN = kTile[1] * kTile[3] * 16, kNT = kTile[0] * kTile[1] * 64;

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 25088, reps = argc > 2 ? atoi(argv[2]) : 5, K = GB_K, N = GB_N;
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  for (auto& v : hA) v = u(rng);
  for (auto& v : hW) v = 0.05f * u(rng);
  for (auto& v : hb) v = 0.1f * u(rng);
  float *A, *W, *Wx, *Y, *bias;
  CK(hipMalloc((void**)&A, hA.size() * 4)); CK(hipMalloc((void**)&W, hW.size() * 4)); CK(hipMalloc((void**)&Wx, hW.size() * 4));
  CK(hipMalloc((void**)&Y, (size_t)M * N * 4)); CK(hipMalloc((void**)&bias, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  const long long groups = (long long)N * K / 32;
  hipLaunchKernelGGL(split_bf16_weights_kernel, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, 0, (const float*)W, Wx, groups);
  CK(hipDeviceSynchronize());
  GemmArgs a;
  a.A = A; a.lda = K; a.K1 = K; a.W = Wx; a.ldw = K; a.w_split = 1; a.bias = bias; a.Y = Y; a.ldy = N; a.M = M; a.N = N;
  constexpr int lds = gemm_lds_bytes<GB_TILE>();
  CK(hipFuncSetAttribute((const void*)GB_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef GB_PIPE
  const dim3 grid(gemm_pipe_grid<GB_TILE>(M, N));
#else
  const dim3 grid((M + kBM - 1) / kBM, (N + kBN - 1) / kBN);
#endif
#ifdef GP_TRACE
  unsigned long long* trace;
  CK(hipMalloc((void**)&trace, (size_t)grid.x * 8 * sizeof(unsigned long long)));
  CK(hipMemset(trace, 0, (size_t)grid.x * 8 * sizeof(unsigned long long)));
  a.trace = trace;
#endif
  if (argc > 3) a.act = atoi(argv[3]);     // 1 = GELU (the host check below then applies it too)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((GB_KERNEL), grid, dim3(kNT), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (r > 1) ms.push_back(t);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  // check rows 0, 1000 and M - 1 against a double-precision product
  double worst = 0, cs = 0;
  for (int row : {0, 1000 < M ? 1000 : 0, M - 1}) {
    std::vector<float> hy(N);
    CK(hipMemcpy(hy.data(), Y + (size_t)row * N, N * 4, hipMemcpyDeviceToHost));
    for (int n = 0; n < N; ++n) {
      double s = hb[n];
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)row * K + k] * hW[(size_t)n * K + k];
      if (a.act == 1) s = 0.5 * s * (1.0 + std::erf(s / std::sqrt(2.0)));
      worst = std::max(worst, std::fabs(s - hy[n]));
      if (row == M - 1) cs += hy[n];
    }
  }
#ifdef GP_TRACE
  {   // per-workgroup phase lengths (shader clocks) of the last launch, averaged; start / end spread on the 100 MHz wall clock
    std::vector<unsigned long long> ht((size_t)grid.x * 8);
    CK(hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost));
    double ph[5] = {0, 0, 0, 0, 0}; int n = 0; unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> starts, ends;
    for (unsigned w = 0; w < grid.x; ++w) {
      const unsigned long long* o = &ht[(size_t)w * 8];
      if (!o[5]) continue;
      for (int i = 0; i < 5; ++i) ph[i] += (double)(o[i + 1] - o[i]);
      ++n; t0 = std::min(t0, o[6]); t1 = std::max(t1, o[7]);
      starts.push_back((double)o[6]); ends.push_back((double)o[7]);
    }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    printf("{\"trace_workgroups\": %d, \"clk_prologue\": %.0f, \"clk_loop\": %.0f, \"clk_tile_to_lds\": %.0f, \"clk_store_issue\": %.0f, \"clk_store_drain\": %.0f, "
           "\"wall_us\": %.2f, \"median_start_us\": %.2f, \"median_end_us\": %.2f}\n", n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, (t1 - t0) / 100.0,
           (starts[starts.size() / 2] - t0) / 100.0, (ends[ends.size() / 2] - t0) / 100.0);
  }
#endif
  const double flop = 2.0 * M * (double)K * N;
  printf("{\"variant\": \"%s\", \"tile\": \"%dx%d\", \"M\": %d, \"K\": %d, \"N\": %d, \"workgroups\": %u, \"lds\": %d, \"us_min\": %.1f, \"us_med\": %.1f, \"tflops\": %.1f, "
         "\"max_err\": %.3g, \"checksum\": %.6g}\n", LB_NAME, kBM, kBN, M, K, N, grid.x * grid.y, lds, ms.front() * 1e3, ms[ms.size() / 2] * 1e3,
         flop / (ms.front() * 1e-3) / 1e12, worst, cs);
  return 0;
}
