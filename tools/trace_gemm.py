#!/usr/bin/env python
"""Phase breakdown of the decoder's staged GEMMs from in-kernel timestamps (run on the GPU box)."""
import os, sys, json
REPO = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

B, T = int(os.environ.get("TRACE_B", "64")), 196
out = {}
for prec in [int(x) for x in os.environ.get("TRACE_PREC", "0,1").split(",")]:
    eng = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=B, max_frames=T, precision=prec)
    eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
    b = syn.make_batch(B)
    dev = torch.device("cuda:0")
    text, lat0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
    j = torch.empty(B, T, 22, 3, device=dev)
    eng.sample(text, lat0, b.lengths, None, None, j); torch.cuda.synchronize()
    for name in ("dec_qkv", "dec_ffn1", "dec_ffn2_ln", "dec_outproj_ln"):
        tr = eng.profile_trace(name, B, T).astype(np.int64)[:, :4, :]      # [wg<512, 4 waves, 8]
        live = tr[:, 0, 0] != 0
        tr = tr[live]
        d = np.diff(tr[:, :, :5], axis=2).astype(np.float64)
        names = ["prologue(first chunk)", "chunks 0-3", "remaining chunks", "epilogue"]
        rt = tr[:, :, 6:8]
        wg_us = (rt[:, :, 1].max(1) - rt[:, :, 0].min(1)) / 100.0
        cyc = (tr[:, :, 4] - tr[:, :, 0]).max(1).astype(np.float64)
        clk = float(np.median(cyc / np.maximum(wg_us, 1e-9)) / 1e3)
        span = (rt[:, :, 1].max() - rt[:, :, 0].min()) / 100.0
        out[f"{name}/prec{prec}"] = {"traced_wgs": int(live.sum()), "first512_span_us": round(float(span), 1),
                                     "wg_duration_us_median": round(float(np.median(wg_us)), 2), "clock_ghz_est": round(clk, 2),
                                     "phase_cycles_median": {n: int(np.median(d[:, :, i])) for i, n in enumerate(names)}}
        print(name, "prec", prec, json.dumps(out[f"{name}/prec{prec}"]))
    eng.close()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "gemm_trace.json"), "w"), indent=1)
