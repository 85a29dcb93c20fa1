"""Worksheet experiment: does the decoder of the 1 280-motion call gain from running as TWO half batches on two streams (the effect that gives
the diffusion-only variant 12 %: one-workgroup-per-CU kernels leave gaps another stream fills)?  mldhip_vae_decode of N motions in one call
against two concurrent calls of N / 2 on two workspaces of the same handle, over candidate stream pairs (ROCm maps streams to 4 hardware
queues; a pair that collides serialises).    python tools/ab_decode_split.py   [AB_N=1280]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn

N = int(os.environ.get("AB_N", "1280"))
dev = torch.device("cuda:0")
e = _lib.Engine(device=0, max_batch=N, max_frames=196, precision=1, max_in_flight=2)
e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
g = torch.Generator(device="cpu").manual_seed(3)
z = (torch.randn(N, 1, 256, generator=g) * 20).to(dev)
lens = [196] * N
feats = torch.empty(N, 196, 263, device=dev)
ref = torch.empty_like(feats)
streams = [torch.cuda.Stream(device=dev) for _ in range(5)]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(min(ts) * 1e3, 3)


res = {"motions": N}
res["one_call_ms"] = timed(lambda: e.vae_decode(z, lens, ref, streams[0].cuda_stream))
h = N // 2
for j in range(1, 5):
    def two():
        e.vae_decode(z[:h], lens[:h], feats[:h], streams[0].cuda_stream)
        e.vae_decode(z[h:], lens[h:], feats[h:], streams[j].cuda_stream)
    res["two_halves_streams_0+%d_ms" % j] = timed(two)
res["one_half_alone_ms"] = timed(lambda: e.vae_decode(z[:h], lens[:h], feats[:h], streams[0].cuda_stream))
res["max_abs_diff_halves_vs_one_call"] = float((feats - ref).abs().max())
print(json.dumps(res))
e.close()
