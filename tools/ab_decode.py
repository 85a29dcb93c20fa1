"""GPU A/B of decoder options at the headline call shape (N motions per mldhip_sample_many call, split precision mode):
decode time = whole call - loop-only call; joints of request 0 against the reference fixture.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
N = int(os.environ.get("AB_N", "2048"))
OPTS = json.loads(os.environ.get("AB_OPTS", '[{"dec_tail": 1}, {"dec_tail": 0}, {"dec_tail": 1}, {"dec_tail": 0}, {"dec_tail": 1, "strip_gemm": 0}, {"strip_gemm": 1, "ffn_strip": 0}, {"ffn_strip": 1}]'))
eng = _lib.Engine(device=0, max_batch=N, max_frames=196, precision=1)
eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
reqs = []
for i in range(N // 64):
    b = syn.make_batch(64) if i == 0 else syn.make_batch(64, None, seed=1234 + i)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     latents_out=torch.zeros(64, 1, 256, device=dev), joints_out=torch.zeros(64, 196, 22, 3, device=dev)))
lat_only = [dict(text_emb=q["text_emb"], init_latents=q["init_latents"], lengths=q["lengths"], latents_out=q["latents_out"]) for q in reqs]
g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_b64.npz"))


def best(fn, n=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


out = {}
t_loop = best(lambda: eng.sample_many(lat_only))
out["loop_ms"] = round(t_loop * 1e3, 2)
for opt in OPTS:
    for k, v in opt.items():
        eng.set_option(k, v)
    t_all = best(lambda: eng.sample_many(reqs))
    err = float(np.abs(reqs[0]["joints_out"].cpu().numpy()[:, ::4] - g["joints_every4"]).max())
    key = ",".join(f"{k}={v}" for k, v in opt.items())
    while key in out:
        key += "#"                                   # a repeated variant (interleaved rounds) keeps its own entry
    out[key] = dict(all_ms=round(t_all * 1e3, 2), decode_ms=round((t_all - t_loop) * 1e3, 2), motions_per_s=round(N / t_all, 1), joints_err=err)
    print(key, out[key], flush=True)
print(json.dumps(out))
