"""Experiment: N engines (handles) on N streams of one GPU, each sampling its own bs-64 batch concurrently."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
res = {}
for nfl in (1, 2, 3):
    engs, streams, bufs = [], [], []
    for i in range(nfl):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196)
        e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
        m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
        b = syn.make_batch(64, None, seed=1234 + i, max_len=196)
        engs.append(e); streams.append(torch.cuda.Stream())
        bufs.append((torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev), b.lengths,
                     torch.empty(64, 196, 22, 3, device=dev)))
    def run(k):
        for _ in range(k):
            for e, st, (t, x, l, j) in zip(engs, streams, bufs):
                e.sample(t, x, l, None, None, j, st.cuda_stream)
    run(2); torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[nfl] = {"motions_per_s": round(64 * nfl * K / dt, 1), "ms_per_round": round(dt / K * 1e3, 3)}
    for e in engs: e.close()
print(json.dumps(res))
