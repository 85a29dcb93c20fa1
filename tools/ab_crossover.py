"""Measured crossover table of the three reverse-loop kernel families (the auto thresholds were constants): loop-only
calls (latents out, no decode) of B motions on the latency kernels (tile32.hpp, "loop_kernel" 1), the column-split throughput kernels
(strip.hpp, 2) and the sample-major persistent loop (loop_fused.hpp, 3), split-f16 mode and exact-fp32 mode, best of 5, one MI355X.
Prints one JSON object; copy it to profiles/r04_loop_crossover.json."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
out = {"unit": "ms per loop-only call (50 steps, CFG 7.5)", "families": {"1": "latency kernels (tile32.hpp)", "2": "column-split throughput kernels (strip.hpp)", "3": "persistent loop (loop_fused.hpp)"}}
for prec, name in ((1, "f16x3"), (0, "f32")):
    sizes = (64, 128, 192, 256, 320, 448, 640) if prec == 1 else (256, 640, 1024, 1280, 1536)
    eng = _lib.Engine(device=0, max_batch=max(sizes), max_frames=196, precision=prec)
    eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
    tab = {}
    for B in sizes:
        b = syn.make_batch(B, None, seed=7)
        text, lat0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
        lat = torch.empty(B, 1, 256, device=dev)
        row = {}
        for lk in (1, 2, 3):
            if lk == 1 and B > 256:
                continue
            eng.set_option("loop_kernel", lk)
            for _ in range(2):
                eng.sample(text, lat0, b.lengths, lat, None, None)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); eng.sample(text, lat0, b.lengths, lat, None, None); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row[str(lk)] = round(min(ts) * 1e3, 2)
        row["best"] = min(row, key=row.get)
        tab[str(B)] = row
        print(name, B, row, flush=True)
    eng.set_option("loop_kernel", 0)
    out[name] = tab
    eng.close()
print(json.dumps(out))
