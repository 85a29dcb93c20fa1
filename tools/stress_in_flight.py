import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "motion-latent-diffusion_amd"))
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
def mk(nfl):
    e = _lib.Engine(device=0, max_batch=32, max_frames=120, max_in_flight=nfl, num_inference_steps=10)
    e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize(); return e
e4, e1 = mk(4), mk(1)
streams = [torch.cuda.Stream() for _ in range(5)]          # 5 streams on 4 contexts: contexts get reused across streams
rng = np.random.default_rng(0)
jobs = []
for i in range(int(os.environ.get("STRESS_JOBS", "40"))):
    B = int(rng.integers(1, 33)); lens = [int(v) for v in rng.integers(1, 121, B)]
    b = syn.make_batch(B, lens, seed=500 + i)
    jobs.append((b, torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev), torch.empty(B, max(lens), 22, 3, device=dev)))
torch.cuda.synchronize()
for rep in range(3):
    order = rng.permutation(len(jobs))
    for k, i in enumerate(order):
        b, t, x, j = jobs[i]
        e4.sample(t, x, b.lengths, None, None, j, streams[int(rng.integers(0, 5))].cuda_stream)
    torch.cuda.synchronize()
bad = 0
for b, t, x, j in jobs:
    ref = torch.empty_like(j); e1.sample(t, x, b.lengths, None, None, ref); torch.cuda.synchronize()
    bad += int(not torch.equal(j, ref))
print("stress: jobs", len(jobs), "mismatches", bad)
