/* Measurement hooks of the mldhip engine -- NOT part of the production ABI.
 *
 * libmldhip.so exports the sampling surface of include/mldhip.h only.  `make -C motion-latent-diffusion_amd/csrc hooks` builds the same sources with
 * -DMLDHIP_HOOKS into mld_hip/libmldhip_hooks.so, which additionally exports the two entry points below, knows the options "fused_dbg", "cluster_graph",
 * "cluster_lane", "cluster_chunk", "cluster_stale" and "cluster_inject" (documented with the production options in mldhip.h) and carries the traced instantiations of the loop
 * kernels.  tools/trace_*.py, tools/ab_coalesce.py, tools/dbg_cluster.py and tools/two_streams.py load that library
 * (mld_hip._lib.hooks_library()), and so does ONE GPU test (fault injection into the cluster loop's bounded waits); bench.py, every other test and the mld_hip
 * package do not. */
#ifndef MLDHIP_HOOKS_H_
#define MLDHIP_HOOKS_H_
#include "mldhip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the declarations of this header are its only exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Measurement hook (no reference counterpart): enqueue ONE named kernel of the path `iters` times on
 * `stream` at its production shape for batch B / Tmax T; writes its algorithmic FLOPs per launch.
 * names: den_{qkv,outproj,ffn1,ffn2,final}, dec_{qkv,attn,outproj_ln,ffn1,ffn2_ln}.  The caller times it with events on `stream`. */
int mldhip_profile_kernel(mldhip_handle* h, const char* name, int32_t B, int32_t T, int32_t iters,
                          double* flops_per_launch, void* stream);

/* Measurement hook: one traced launch of a den_* kernel; writes 8 uint64 timestamps per wave (64 per
 * workgroup) to out_host: start, loads landed, LDS written, barrier passed, MFMAs done, stores drained
 * (shader clock) and start/end on the 100 MHz realtime counter.  Returns workgroup slots copied. */
int mldhip_profile_trace(mldhip_handle* h, const char* name, int32_t B, int32_t T, uint64_t* out_host,
                         int64_t cap_u64, void* stream);

/* mldhip_profile_trace also answers three read-back names: "den_loop_phases" (the persistent loop's phase counters under "fused_dbg" 5),
 * "den_cluster_xbuf" (B = cluster index: the exchange region the last cluster-loop call left, kernels/loop_cluster.hpp) and "den_cluster_status" (the status
 * words of the last cluster-loop call: [0] a wait timed out, [1] a cluster spanned XCDs, [2] the sticky timeout word). */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MLDHIP_HOOKS_H_ */
