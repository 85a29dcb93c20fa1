"""Experiment: N handles x N host threads x N streams (one bs-64 chain each) vs one thread rotating over N streams."""
import json, os, sys, time, threading
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
res = {}
for nfl in (4, 5, 6, 8):
    engs, streams, bufs = [], [], []
    for i in range(nfl):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196)
        e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
        m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
        b = syn.make_batch(64, None, seed=1234 + i, max_len=196)
        engs.append(e); streams.append(torch.cuda.Stream())
        bufs.append((torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev), b.lengths,
                     torch.empty(64, 196, 22, 3, device=dev)))
    torch.cuda.synchronize()
    def worker(i, k):
        t, x, l, j = bufs[i]
        for _ in range(k):
            engs[i].sample(t, x, l, None, None, j, streams[i].cuda_stream)
    def run(k):
        th = [threading.Thread(target=worker, args=(i, k)) for i in range(nfl)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(2)
    K = 8
    dt = run(K)
    res[nfl] = {"motions_per_s": round(64 * nfl * K / dt, 1), "ms_per_step_amortised": round(dt / (K * nfl) * 1e3, 3)}
    for e in engs: e.close()
print(json.dumps(res))
