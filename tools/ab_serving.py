"""GPU A/B of serving shapes for the headline workload (bs-64 requests, T=196): NREQ requests coalesced per mldhip_sample_many
call x NFL calls in flight, per arithmetic mode.  GPU_MAX_HW_QUEUES must be set by the caller (bench.py sets 8)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
sd = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()}, **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
sd["mean"], sd["std"] = syn.make_mean_std()
reqs = []
for i in range(16):
    b = syn.make_batch(64, None, seed=1234 + i, max_len=196)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     joints_out=torch.empty(64, 196, 22, 3, device=dev)))
out = {}
for prec, pname in ((0, "f32"), (1, "f16x3")):
    for nreq in (1, 2, 4, 8):
        for nfl in (1, 2, 3, 4):
            if nreq * nfl > 16: continue
            eng = _lib.Engine(device=0, max_batch=64 * nreq, max_frames=196, max_in_flight=nfl, precision=prec)
            eng.load_state_dict(sd); eng.finalize()
            for opt in ((0, 2) if nreq * 384 >= 768 else (0,)):
                eng.set_option("loop_kernel", opt)
                streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
                call = lambda i: eng.sample_many(reqs[(i % nfl) * nreq:(i % nfl + 1) * nreq], streams[i % nfl].cuda_stream)
                for i in range(2 * nfl): call(i)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n = max(3 * nfl, 24 // nreq)
                for i in range(n): call(i)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                key = "%s req%d x fl%d%s" % (pname, nreq, nfl, " (latency kernels forced)" if False else (" auto" if opt == 0 else " strip"))
                out[key] = round(64 * nreq * n / dt, 1)
            eng.close()
print(json.dumps(out, indent=0))
