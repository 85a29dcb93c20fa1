"""GPU A/B of the sample-major persistent loop (kernels/loop_fused.hpp, loop_kernel = 3) against the launch-per-GEMM families.

(1) parity: request 0 = the reference-generated bs-64 / T-196 fixture inside coalesced calls of 64 .. NMAX motions, every loop family;
    all other requests of the biggest call against the same requests through the latency kernels (one bs-64 call each).
(2) time: loop only (latents out) and whole path (joints out), one call at a time, per family and motions per call; the
    column-split families additionally with 4 calls in flight (the round-2 headline shape).
Writes gpurun_out/ab_fused.json."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
NMAX = int(os.environ.get("AB_NMAX", "2048"))
PREC = int(os.environ.get("AB_PREC", "1"))
SIZES = [int(x) for x in os.environ.get("AB_SIZES", "64,320,1024,2048").split(",")]
out = {"prec": PREC, "nmax": NMAX}


def load(eng):
    eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser.")
    eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std()
    eng.load_tensor("mean", m); eng.load_tensor("std", s)
    eng.finalize()


eng = _lib.Engine(device=0, max_batch=NMAX, max_frames=196, precision=PREC, max_in_flight=4)
load(eng)
nreq = NMAX // 64
reqs = []
for i in range(nreq):
    b = syn.make_batch(64) if i == 0 else syn.make_batch(64, None, seed=1234 + i, max_len=196)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     joints_out=torch.zeros(64, 196, 22, 3, device=dev), latents_out=torch.zeros(64, 1, 256, device=dev)))
lat_only = [dict(text_emb=q["text_emb"], init_latents=q["init_latents"], lengths=q["lengths"], latents_out=q["latents_out"]) for q in reqs]
g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_b64.npz"))


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


# ---- parity of request 0 (reference fixture) per family and call size
par = {}
FAMS = [(1, 0, "fam1"), (2, 0, "fam2"), (3, 0, "fam3"), (3, 1, "fam3x3")] if PREC == 1 else [(1, 0, "fam1"), (2, 0, "fam2"), (3, 0, "fam3")]
for fam, x3, tag in FAMS:
    eng.set_option("loop_kernel", fam)
    eng.set_option("fused_x3", x3)
    for n in SIZES:
        if n > NMAX or (fam == 1 and n > 320):
            continue
        for q in reqs[:n // 64]:
            q["joints_out"].zero_(); q["latents_out"].zero_()
        eng.sample_many(reqs[:n // 64]); torch.cuda.synchronize()
        q = reqs[0]
        par[f"{tag}_B{n}"] = dict(latents=float(np.abs(q["latents_out"].cpu().numpy() - g["latents"]).max()),
                                  joints=float(np.abs(q["joints_out"].cpu().numpy()[:, ::4] - g["joints_every4"]).max()))
        print("parity", tag, n, par[f"{tag}_B{n}"], flush=True)
out["parity_request0_vs_reference_fixture"] = par

# ---- every request of the biggest fused call vs the same request alone on the latency kernels
eng.set_option("loop_kernel", 3)
eng.set_option("fused_x3", 1 if PREC == 1 else 0)
eng.sample_many(reqs); torch.cuda.synchronize()
fused_j = [q["joints_out"].clone() for q in reqs]
fused_l = [q["latents_out"].clone() for q in reqs]
eng.set_option("loop_kernel", 1)
worst = dict(joints=0.0, latents=0.0)
per = []
for i, q in enumerate(reqs):
    eng.sample_many([q]); torch.cuda.synchronize()
    ej = float((q["joints_out"] - fused_j[i]).abs().max()); el = float((q["latents_out"] - fused_l[i]).abs().max())
    per.append((round(el, 6), round(ej, 6)))
    worst["joints"] = max(worst["joints"], ej); worst["latents"] = max(worst["latents"], el)
out["fused_call_vs_single_requests_on_latency_kernels"] = dict(worst=worst, per_request_latents_joints=per)
print("fused vs latency kernels, all requests:", worst, flush=True)

# ---- timing, one call at a time
tim = {}
for fam, x3, tag in FAMS[1:]:
    eng.set_option("loop_kernel", fam)
    eng.set_option("fused_x3", x3)
    for n in SIZES:
        if n > NMAX:
            continue
        k = n // 64
        t_loop = timed(lambda: eng.sample_many(lat_only[:k]))
        t_all = timed(lambda: eng.sample_many(reqs[:k]))
        tim[f"{tag}_B{n}"] = dict(loop_ms=round(t_loop * 1e3, 2), all_ms=round(t_all * 1e3, 2), motions_per_s=round(n / t_all, 1),
                                     loop_motions_per_s=round(n / t_loop, 1))
        print("time", tag, n, tim[f"{tag}_B{n}"], flush=True)
out["one_call_at_a_time"] = tim

# ---- calls in flight (4 streams), 320 motions per call for the column-split family; fused: 2 x 1024
def in_flight(fam, per_call, nfl, ncalls, x3=0):
    eng.set_option("loop_kernel", fam)
    eng.set_option("fused_x3", x3)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    k = per_call // 64
    groups = [reqs[(i * k) % nreq:(i * k) % nreq + k] for i in range(nfl)]
    for i in range(nfl):
        eng.sample_many(groups[i], streams[i].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(ncalls):
        eng.sample_many(groups[i % nfl], streams[i % nfl].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return round(per_call * ncalls / dt, 1)


fl = {}
fl["fam2_320x4"] = in_flight(2, 320, 4, 8)
if NMAX >= 2048:
    x3 = 1 if PREC == 1 else 0
    fl["fam3_1024x2"] = in_flight(3, 1024, 2, 4, x3)
    fl["fam3_2048x1"] = in_flight(3, 2048, 1, 2, x3)
    fl["fam3_2048x2"] = in_flight(3, 2048, 2, 4, x3)
print("in flight:", fl, flush=True)
out["in_flight_motions_per_s"] = fl
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_fused.json"), "w"), indent=1)
print(json.dumps(out)[:3000])
