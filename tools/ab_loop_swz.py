"""GPU A/B of "fused_swz" (row-swizzled operand images of the persistent loop) at the headline call shape: loop-only calls
(latents out, no decode) of N motions, interleaved rounds, best-of-4 wall time per variant and round; latents of both forms must
agree to the bit.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
N = int(os.environ.get("AB_N", "2048"))
eng = _lib.Engine(device=0, max_batch=N, max_frames=196, precision=1)
eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
reqs = []
for i in range(N // 64):
    b = syn.make_batch(64) if i == 0 else syn.make_batch(64, None, seed=1234 + i)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     latents_out=torch.zeros(64, 1, 256, device=dev)))


def best(fn, n=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


out, lat = {"rounds": []}, {}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    row = {}
    for swz in (0, 1):
        eng.set_option("fused_swz", swz)
        row["swz%d_ms" % swz] = round(best(lambda: eng.sample_many(reqs)) * 1e3, 3)
        lat[swz] = torch.cat([q["latents_out"] for q in reqs]).clone()
    out["rounds"].append(row)
    print(row, flush=True)
out["latents_bit_identical"] = bool(torch.equal(lat[0], lat[1]))
out["latents_max_abs_diff"] = float((lat[0] - lat[1]).abs().max().item())
out["swz0_ms_min"] = min(r["swz0_ms"] for r in out["rounds"]); out["swz1_ms_min"] = min(r["swz1_ms"] for r in out["rounds"])
print(json.dumps(out))
