// Stand-alone timing harness for the cluster form of the reverse loop (kernels/loop_cluster.hpp): the kernel alone on random weight
// fragments of the right shape (9 layers, 50 steps), without the engine around it -- built in seconds:
//   LB_SRC=cluster_bench.hip tools/loopbench/build.sh cluster_bench [-DCB_WT=false]      run:  build/lb/cluster_bench [motions=64] [reps=5] [steps=50]
// Prints one JSON line: ms per launch (min / median), the status word, a checksum of the latents (equal between variants that compute the
// same thing).  Timing only -- results are checked through the library (tests/, simulator + GPU), never here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "loop_cluster.hpp"

#ifndef CB_WT
#define CB_WT true
#endif
#ifndef CB_CG
#define CB_CG 4
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using namespace mld;

// stand-in for the decoder that runs between two reverse loops of consecutive full calls: dirties `n` floats (GBs through every L2)
__global__ void dirty_kernel(float* __restrict__ p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 5, n = argc > 3 ? atoi(argv[3]) : 50;
  const int L = 9, nb = 4, ncl = (B + 7) / 8;
  if (ncl > kClMaxClusters) { fprintf(stderr, "at most %d motions\n", 8 * kClMaxClusters); return 1; }
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  auto dev = [&](size_t nfl, float scale, float bias, float** out) {
    std::vector<float> h(nfl);
    for (auto& v : h) v = bias + scale * u(rng);
    if (hipMalloc((void**)out, nfl * sizeof(float)) != hipSuccess) return 1;
    return hipMemcpy(*out, h.data(), nfl * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  // per (column group, wave): fragments per step as den_cluster_kernel consumes them (4 groups: waves 0-3 52 per layer, waves 4-7 44, + 8 per skip linear;
  // 8 groups: 20 / 12 in Ph1 for groups 0-3, 16 in the feed-forward block, + 4 per skip linear)
  ClusterArgs a;
  size_t nfrag = 0;
  unsigned woff[CB_CG * 8];
  auto per_step = [&](int hc, int w) {
    if (CB_CG == 4) return (size_t)L * (w < 4 ? 52 : 44) + (size_t)nb * 8;
    return (size_t)L * ((hc < 4 ? (w < 4 ? 20 : 12) : 0) + 16) + (size_t)nb * 4;
  };
  for (int hc = 0; hc < CB_CG; ++hc)
    for (int w = 0; w < 8; ++w) {
      woff[hc * 8 + w] = (unsigned)(nfrag * kClFragFloats);
      nfrag += per_step(hc, w) + kClRing;
    }
  unsigned* woff_dev;
  CK(hipMalloc((void**)&woff_dev, sizeof woff));
  CK(hipMemcpy(woff_dev, woff, sizeof woff, hipMemcpyHostToDevice));
  a.wave_off = woff_dev;
  float *arena, *stream, *small, *T1, *TP, *init, *lat, *park, *ddim, *xbuf;
  unsigned* flags;
  if (dev(nfrag * 16 * 32, 0.06f, 0.f, &arena)) return 1;
  CK(hipMalloc((void**)&stream, nfrag * kClFragFloats * sizeof(float)));
  {
    std::vector<ClFrag> fr(nfrag);
    size_t i = 0;
    for (int hc = 0; hc < CB_CG; ++hc)
      for (int w = 0; w < 8; ++w) {
        const size_t per = per_step(hc, w), first = i;
        for (size_t k = 0; k < per; ++k, ++i) fr[i] = ClFrag{(long long)i * 512, 32, 0};
        for (int k = 0; k < kClRing; ++k, ++i) fr[i] = fr[first + k];
      }
    ClFrag* fdev;
    CK(hipMalloc((void**)&fdev, nfrag * sizeof(ClFrag)));
    CK(hipMemcpy(fdev, fr.data(), nfrag * sizeof(ClFrag), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_cluster_frags_kernel, dim3((unsigned)nfrag), dim3(64), 0, 0, (const float*)arena, (const ClFrag*)fdev, stream);
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
  const int grid = 8 * 3 * CB_CG * ((ncl + 7) / 8);
  CK(hipMalloc((void**)&lat, (size_t)B * 256 * sizeof(float)));
  CK(hipMalloc((void**)&park, (size_t)grid * nb * 16 * 256 * sizeof(float)));
  CK(hipMalloc((void**)&xbuf, (size_t)ncl * kClXFloats * sizeof(float)));
  CK(hipMalloc((void**)&flags, ((size_t)ncl * kClFlagWords + 16) * sizeof(unsigned)));
  {
    std::vector<float> h((size_t)n * 4);
    for (int s = 0; s < n; ++s) { h[4 * s] = 0.9f; h[4 * s + 1] = 0.43f; h[4 * s + 2] = 0.92f; h[4 * s + 3] = 0.39f; }
    CK(hipMalloc((void**)&ddim, h.size() * sizeof(float)));
    CK(hipMemcpy(ddim, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  a.stream = stream; a.small = small; a.T1 = T1; a.TP = TP; a.init_lat = init; a.lat = lat; a.park = park; a.ddim = ddim; a.xbuf = xbuf;
  a.flags = flags; a.status = flags + (size_t)ncl * kClFlagWords; a.B = B; a.s_base = 0; a.s_end = B; a.timeout = kClTimeoutTicks; a.L = L; a.n = n; a.ncl = ncl; a.xslots = 8; a.guidance = 7.5f; a.init_sigma = 1.f;
  CK(hipFuncSetAttribute((const void*)den_cluster_kernel<CB_WT, CB_CG>, hipFuncAttributeMaxDynamicSharedMemorySize, kClLdsBytes));
#ifdef CL_TRACE
  unsigned long long* tr;
  CK(hipMalloc((void**)&tr, (size_t)grid * 8 * 16 * sizeof(unsigned long long)));
  CK(hipMemset(tr, 0, (size_t)grid * 8 * 16 * sizeof(unsigned long long)));
  a.trace = tr;
#endif
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  const size_t dirty_floats = argc > 4 ? (size_t)atoll(argv[4]) << 20 : 0;      // MiB of floats written between launches (0: none)
  float* dirty = nullptr;
  if (dirty_floats) CK(hipMalloc((void**)&dirty, dirty_floats * sizeof(float)));
  std::vector<double> sums;
  for (int it = 0; it < reps + 1; ++it) {
    if (dirty) hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, dirty, dirty_floats, (float)it);
    hipLaunchKernelGGL(clear_cluster_flags_kernel, dim3(1), dim3(256), 0, 0, flags, (int)(ncl * kClFlagWords + 16));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((den_cluster_kernel<CB_WT, CB_CG>), dim3(grid), dim3(512), kClLdsBytes, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float t = 0.f;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (it) ms.push_back(t);
    {
      std::vector<float> hh((size_t)B * 256);
      CK(hipMemcpy(hh.data(), lat, hh.size() * sizeof(float), hipMemcpyDeviceToHost));
      double c2 = 0;
      for (float v : hh) c2 += v;
      sums.push_back(c2);
    }
  }
  printf("{\"checksums_per_launch\": [");
  for (size_t i = 0; i < sums.size(); ++i) printf("%s%.6f", i ? ", " : "", sums[i]);
  printf("]}\n");
  std::sort(ms.begin(), ms.end());
  std::vector<float> h((size_t)B * 256);
  unsigned st[2] = {0, 0};
  CK(hipMemcpy(h.data(), lat, h.size() * sizeof(float), hipMemcpyDeviceToHost));
  CK(hipMemcpy(st, a.status, sizeof st, hipMemcpyDeviceToHost));
  double cs = 0;
  for (float v : h) cs += v;
#ifdef CL_TRACE
  {
    std::vector<unsigned long long> ht((size_t)grid * 8 * 16);
    CK(hipMemcpy(ht.data(), tr, ht.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static const char* names[16] = {"ph1_qkv_attn", "publish_ao", "wait_e1", "gather_e1", "norm1", "linear1_gelu", "publish_h", "wait_e2", "gather_e2", "linear2_reduce_store", "publish_y", "wait_e3", "e3_rest", "outproj_products", "e3_loads_normal_layers", "linear2_products"};
    for (int wv = 0; wv < 8; wv += 4) {
      double tot = 0, sum[16] = {0};
      int cnt = 0;
      for (int b = 0; b < grid; ++b) {
        const unsigned long long* o = ht.data() + ((size_t)b * 8 + wv) * 16;
        double tb = 0;
        for (int k = 0; k < 16; ++k) tb += (double)o[k];
        if (tb == 0) continue;
        ++cnt;
        for (int k = 0; k < 16; ++k) sum[k] += (double)o[k];
        tot += tb;
      }
      printf("{\"trace_wave\": %d, \"workgroups\": %d, \"cycles_per_layer\": %.0f, \"share\": {", wv, cnt, tot / cnt / (n * L));
      for (int k = 0; k < 16; ++k) printf("%s\"%s\": %.0f", k ? ", " : "", names[k], sum[k] / cnt / (n * L));
      printf("}}\n");
    }
  }
#endif
  printf("{\"name\": \"%s\", \"motions\": %d, \"clusters\": %d, \"workgroups\": %d, \"steps\": %d, \"column_groups\": %d, \"write_through\": %d, \"ms_min\": %.3f, \"ms_median\": %.3f, \"us_per_layer\": %.2f, \"status\": %u, \"latents_checksum\": %.6f}\n",
         LB_NAME, B, ncl, grid, n, CB_CG, (int)CB_WT, ms.front(), ms[ms.size() / 2], ms[ms.size() / 2] * 1000.0 / (n * L), st[0], cs);
  return 0;
}
