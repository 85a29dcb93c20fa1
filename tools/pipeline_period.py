"""(GPU, round 6) Per-request period and per-call overhead of the pipelined bs-64 mode ("many_pipeline"): wall time of calls of n = 1, 2, 5, 10, 20, 40 requests
(barrier-free, one process: synchronize on both sides, best and median of 7), and the least-squares line t(n) = overhead + n x period through them."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1, max_in_flight=2)
e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
reqs = []
for i in range(40):
    b = syn.make_batch(64, None, seed=500 + i % 4)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     latents_out=torch.empty(64, 1, 256, device=dev), joints_out=torch.empty(64, 196, 22, 3, device=dev)))
st = torch.cuda.Stream()
out = {}
for mode in ("serial", "pipelined"):
    e.set_option("many_pipeline", 1 if mode == "pipelined" else 0)
    rows = {}
    for n in (1, 2, 5, 10, 20, 40):
        def call():
            if mode == "serial" or n == 1:
                for q in reqs[:n]:
                    e.sample(q["text_emb"], q["init_latents"], q["lengths"], q["latents_out"], None, q["joints_out"], st.cuda_stream)
            else:
                e.sample_many(reqs[:n], st.cuda_stream)
        call(); st.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); call(); st.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        rows[n] = {"ms_min": round(min(ts), 3), "ms_median": round(float(np.median(ts)), 3)}
    ns = np.array(sorted(rows)); tm = np.array([rows[n]["ms_median"] for n in ns])
    A = np.stack([np.ones_like(ns, dtype=float), ns.astype(float)], 1)
    (ov, per), *_ = np.linalg.lstsq(A[1:], tm[1:], rcond=None)          # (n = 1 is a plain mldhip_sample call in both modes)
    out[mode] = {"calls": {str(k): v for k, v in rows.items()}, "fit_over_n_ge_2": {"overhead_ms_per_call": round(float(ov), 3), "period_ms_per_request": round(float(per), 4)}}
print(json.dumps(out))
