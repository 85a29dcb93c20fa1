#!/usr/bin/env python
"""Phase breakdown of the throughput (strip.hpp) denoiser kernels from in-kernel timestamps, at the headline call shape
(5 x 64 motions -> 1 920 token rows).  Run on the GPU box; writes gpurun_out/strip_trace.json."""
import os, sys, json
REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

B, T = int(os.environ.get("TRACE_B", "320")), 196
eng = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=B, max_frames=T, precision=1)
eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
torch.cuda.synchronize()
out = {"rows": 6 * B}
names = ["loads_landed+prologue", "lds_write+barrier", "chunks_first_half", "chunks_second_half", "epilogue"]
for name in ("den_qkv", "den_outproj", "den_ffn1"):
    for rep in range(2):
        tr = eng.profile_trace(name, B, T).astype(np.int64)          # [wg, wave, 8]
    live = tr[:, :, 0] != 0
    nwg = int(live.any(1).sum())
    tr = tr[:nwg]
    d = np.diff(tr[:, :, :6], axis=2).astype(np.float64)          # phases in shader cycles
    rt = tr[:, :, 6:8]
    span_us = (rt[:, :, 1].max() - rt[:, :, 0].min()) / 100.0     # 100 MHz realtime counter
    start_us = (rt[:, :, 0].min(1) - rt[:, :, 0].min()) / 100.0   # per-workgroup start offset
    wg_us = ((rt[:, :, 1].max(1) - rt[:, :, 0].min(1)) / 100.0)
    total_cyc = (tr[:, :, 5] - tr[:, :, 0]).astype(np.float64)
    clk_ghz = float(np.median(total_cyc.max(1) / np.maximum(wg_us, 1e-9)) / 1e3)
    out[name] = {"workgroups": nwg, "kernel_span_us": round(span_us, 2),
                 "wg_start_offset_us": {"median": round(float(np.median(start_us)), 2), "p90": round(float(np.percentile(start_us, 90)), 2), "max": round(float(start_us.max()), 2)},
                 "wg_duration_us": {"median": round(float(np.median(wg_us)), 2), "p10": round(float(np.percentile(wg_us, 10)), 2), "max": round(float(wg_us.max()), 2)},
                 "shader_clock_ghz_est": round(clk_ghz, 2),
                 "phase_cycles_median": {n: int(np.median(d[:, :, i])) for i, n in enumerate(names)},
                 "phase_cycles_p90": {n: int(np.percentile(d[:, :, i], 90)) for i, n in enumerate(names)},
                 "phase_cycles_p10": {n: int(np.percentile(d[:, :, i], 10)) for i, n in enumerate(names)}}
    print(name, json.dumps(out[name]))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "strip_trace.json"), "w"), indent=1)
