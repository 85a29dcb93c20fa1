#!/usr/bin/env python
"""Phase breakdown of the denoiser tile32 kernels from in-kernel timestamps (run on the GPU box)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

B, T = 64, 196
PREC = int(os.environ.get("TRACE_PREC", "1"))                  # 1 = split-f16 (the latency kernels on split-f16 MFMAs), 0 = exact fp32
eng = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=B, max_frames=T, precision=PREC)
eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
b = syn.make_batch(B)
dev = torch.device("cuda:0")
text, lat0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
j = torch.empty(B, T, 22, 3, device=dev)
eng.sample(text, lat0, b.lengths, None, None, j); torch.cuda.synchronize()
out = {}
for name in ("den_qkv", "den_outproj", "den_ffn1", "den_ffn2"):
    tr = eng.profile_trace(name, B, T).astype(np.int64)          # [wg, wave, 8]
    live = tr[:, :, 0] != 0
    nwg = int(live.any(1).sum())
    tr = tr[:nwg]
    d = np.diff(tr[:, :, :6], axis=2).astype(np.float64)          # phases in shader cycles
    names = ["load+prologue", "lds_write", "barrier_wait", "mfma", "epilogue"]
    rt = tr[:, :, 6:8]
    span_us = (rt[:, :, 1].max() - rt[:, :, 0].min()) / 100.0     # 100 MHz realtime counter
    start_skew_us = (rt[:, :, 0].max() - rt[:, :, 0].min()) / 100.0
    wg_us = ((rt[:, :, 1].max(1) - rt[:, :, 0].min(1)) / 100.0)
    total_cyc = (tr[:, :, 5] - tr[:, :, 0]).astype(np.float64)
    clk_ghz = float(np.median(total_cyc.max(1) / np.maximum(wg_us, 1e-9)) / 1e3)
    out[name] = {"workgroups": nwg, "kernel_span_us": round(span_us, 2), "start_skew_us": round(start_skew_us, 2),
                 "wg_duration_us_median": round(float(np.median(wg_us)), 2), "wg_duration_us_max": round(float(wg_us.max()), 2),
                 "shader_clock_ghz_est": round(clk_ghz, 2),
                 "phase_cycles_median": {n: int(np.median(d[:, :, i])) for i, n in enumerate(names)},
                 "phase_cycles_p90": {n: int(np.percentile(d[:, :, i], 90)) for i, n in enumerate(names)}}
    print(name, json.dumps(out[name]))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "tile32_trace_prec%d.json" % PREC), "w"), indent=1)
