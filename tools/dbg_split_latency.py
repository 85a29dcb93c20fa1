"""Latency experiment: one bs-64 request as NSPLIT concurrent sub-batches on NSPLIT streams vs one call (decode included in each)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
sd = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()}, **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
sd["mean"], sd["std"] = syn.make_mean_std()
eng = _lib.Engine(device=0, max_batch=64, max_frames=196, max_in_flight=4, precision=1)
eng.load_state_dict(sd); eng.finalize()
b = syn.make_batch(64, None, seed=1234, max_len=196)
text, lat0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
joints = torch.empty(64, 196, 22, 3, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
out = {}
def run(nsplit, n=12):
    Bc = 64 // nsplit
    parts = []
    for k in range(nsplit):
        t = torch.cat([text[k * Bc:(k + 1) * Bc], text[64 + k * Bc:64 + (k + 1) * Bc]]).contiguous()
        parts.append((t, lat0[k * Bc:(k + 1) * Bc].contiguous(), b.lengths[k * Bc:(k + 1) * Bc], joints[k * Bc:(k + 1) * Bc]))
    def once():
        for k, (t, x, l, j) in enumerate(parts):
            eng.sample(t, x, l, None, None, j, streams[k].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(3): once()
    t0 = time.perf_counter()
    for _ in range(n): once()
    return round((time.perf_counter() - t0) / n * 1e3, 3)
for ns in (1, 2, 4):
    out["ms_per_bs64_request_split_%d" % ns] = run(ns)
print(json.dumps(out))
