"""GPU A/B of per-handle options (mldhip_set_option) at the headline call shape (5 x 64 motions per call, 4 calls in flight):
end-to-end motions/s plus back-to-back per-kernel times.  Environment: AB_NREQ / AB_NFL (requests per call / calls in flight),
AB_DEFAULTS and AB_CONFIGS (JSON: the baseline option values and the list of overrides to run; the first and last entries should be
the baseline, the first one absorbs the clock ramp), AB_KERNELS (comma list of mldhip_profile_kernel names to time)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
NREQ, NFL = int(os.environ.get("AB_NREQ", "5")), int(os.environ.get("AB_NFL", "4"))
sd = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()}, **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
sd["mean"], sd["std"] = syn.make_mean_std()
reqs = []
for i in range(NREQ * NFL):
    b = syn.make_batch(64, None, seed=1234 + i, max_len=196)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     joints_out=torch.empty(64, 196, 22, 3, device=dev)))
eng = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=64 * NREQ, max_frames=196, max_in_flight=NFL, precision=1)
eng.load_state_dict(sd); eng.finalize()
streams = [torch.cuda.Stream(device=dev) for _ in range(NFL)]
stream = torch.cuda.current_stream()
def tk(name, iters=60):
    eng.profile_kernel(name, 64 * NREQ, 196, 3, stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); fl = eng.profile_kernel(name, 64 * NREQ, 196, iters, stream.cuda_stream); e1.record(stream); e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return [round(ms * 1e3, 2), round(fl / ms / 1e9, 1)]
out = {}
ref = None
DEFAULTS = json.loads(os.environ.get("AB_DEFAULTS", '{"strip_wide": 0, "strip_ffn2_split": 2, "strip_waves": 8}'))
CONFIGS = [dict(), dict(strip_wide=1), dict(strip_wide=2), dict(strip_waves=4), dict(strip_wide=1, strip_waves=4), dict(strip_ffn2_split=1), dict()]
if os.environ.get("AB_CONFIGS"): CONFIGS = json.loads(os.environ["AB_CONFIGS"])
for n, cfg in enumerate(CONFIGS):
    opts = {**DEFAULTS, **cfg}
    for k, v in opts.items(): eng.set_option(k, v)
    call = lambda i: eng.sample_many(reqs[(i % NFL) * NREQ:(i % NFL + 1) * NREQ], streams[i % NFL].cuda_stream)
    for i in range(2 * NFL): call(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_calls = 3 * NFL
    for i in range(n_calls): call(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    j = reqs[0]["joints_out"].clone()
    if ref is None: ref = j
    name = "%d: " % n + (" ".join("%s=%d" % (k.replace("strip_", ""), v) for k, v in cfg.items()) or "defaults")
    out[name] = {"motions_per_s": round(64 * NREQ * n_calls / dt, 1), "max_abs_vs_first": float((j - ref).abs().max()),
                 "us_gflops": {k: tk(k) for k in os.environ.get("AB_KERNELS", "den_qkv,den_outproj,den_ffn1,den_ffn2").split(",")}}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps({"requests_per_call": NREQ, "calls_in_flight": NFL, "results": out}, indent=0))
