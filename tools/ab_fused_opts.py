"""GPU: loop-only time of the sample-major persistent loop (loop_kernel = 3) over its options (operand format x ring depth) at one
call size, plus latents of request 0 against the reference fixture.  Prints one JSON line; ~10 s."""
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
g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_b64.npz"))
eng.set_option("loop_kernel", 3)
out = {}
for x3 in (0, 1):
    for ring in (4, 8):
        eng.set_option("fused_x3", x3); eng.set_option("fused_ring", ring)
        eng.sample_many(reqs); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); eng.sample_many(reqs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        err = float(np.abs(reqs[0]["latents_out"].cpu().numpy() - g["latents"]).max())
        out[f"x3={x3},ring={ring}"] = dict(loop_ms=round(min(ts) * 1e3, 2), latents_err=err)
        print(f"x3={x3} ring={ring}", out[f"x3={x3},ring={ring}"], flush=True)
eng.set_option("fused_x3", 1); eng.set_option("fused_ring", 4)
for dbg, what in ((1, "no_weight_stream"), (2, "no_mfma"), (3, "identity_for_gelu"), (4, "no_ffn_epilogue")):
    eng.set_option("fused_dbg", dbg)
    eng.sample_many(reqs); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.sample_many(reqs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out["x3_ring4_" + what] = round(min(ts) * 1e3, 2)
    print(what, out["x3_ring4_" + what], flush=True)
eng.set_option("fused_dbg", 0)
print(json.dumps(out))
