"""GPU A/B: (1) per-kernel time of the loop GEMMs at B = 64 (latency kernels) and B = 256 (throughput kernels, and the
latency kernels forced to the same rows); (2) whole-path throughput: 4 bs-64 requests as 4 calls in flight vs ONE
mldhip_sample_many call (1 and 2 such calls in flight); (3) the coalesced result vs per-request results."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
NREQ = int(os.environ.get("AB_NREQ", "4"))
STEPS = int(os.environ.get("AB_STEPS", "8"))


def load(eng):
    eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser.")
    eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std()
    eng.load_tensor("mean", m); eng.load_tensor("std", s)
    eng.finalize()


def tk(eng, name, B, T, iters, stream):
    eng.profile_kernel(name, B, T, 3, stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); fl = eng.profile_kernel(name, B, T, iters, stream.cuda_stream); e1.record(stream); e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return round(ms * 1e3, 2), round(fl / ms / 1e9, 1)


out = {}
stream = torch.cuda.current_stream()
big = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=64 * NREQ, max_frames=196, max_in_flight=2)
load(big)
reqs = []
for i in range(2 * NREQ):
    b = syn.make_batch(64, None, seed=1234 + i, max_len=196)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     joints_out=torch.empty(64, 196, 22, 3, device=dev), latents_out=torch.empty(64, 1, 256, device=dev)))
big.sample_many(reqs[:NREQ]); torch.cuda.synchronize()
kern = {}
for fam, opt in (("throughput", 2), ("latency", 1)):
    big.set_option("loop_kernel", opt)
    big.sample_many(reqs[:NREQ]); torch.cuda.synchronize()
    kern[fam + "@B%d" % (64 * NREQ)] = {n: tk(big, n, 64 * NREQ, 196, 100, stream) for n in ("den_qkv", "den_outproj", "den_ffn1", "den_ffn2", "den_final")}
big.set_option("loop_kernel", 0)
out["kernels_us_gflops"] = kern

# whole path
def run_many(groups, nsteps):
    streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
    for g in range(groups):
        big.sample_many(reqs[g * NREQ:(g + 1) * NREQ], streams[g].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nsteps):
        g = i % groups
        big.sample_many(reqs[g * NREQ:(g + 1) * NREQ], streams[g].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return round(64 * NREQ * nsteps / dt, 1), round(dt / nsteps * 1e3, 3)

out["coalesced_1_in_flight"] = run_many(1, STEPS)
out["coalesced_2_in_flight"] = run_many(2, STEPS)
big.set_option("loop_kernel", 1)
out["coalesced_latency_kernels"] = run_many(1, max(2, STEPS // 2))
big.set_option("loop_kernel", 0)

small = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=64, max_frames=196, max_in_flight=4)
load(small)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
solo = [torch.empty(64, 196, 22, 3, device=dev) for _ in range(NREQ)]
def run_solo(nsteps, nfl):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nsteps):
        q = reqs[i % NREQ]
        small.sample(q["text_emb"], q["init_latents"], q["lengths"], None, None, solo[i % NREQ], streams[i % nfl].cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return round(64 * nsteps / dt, 1), round(dt / nsteps * 1e3, 3)
run_solo(8, 4)
out["per_request_4_in_flight"] = run_solo(4 * STEPS, 4)
out["per_request_1_in_flight"] = run_solo(2 * STEPS, 1)
small.set_option("loop_kernel", 2)
run_solo(8, 4)
out["per_request_4_in_flight_throughput_kernels"] = run_solo(4 * STEPS, 4)
out["per_request_1_in_flight_throughput_kernels"] = run_solo(2 * STEPS, 1)
out["kernels_us_gflops"]["throughput@B64"] = {n: tk(small, n, 64, 196, 200, stream) for n in ("den_qkv", "den_outproj", "den_ffn1", "den_ffn2", "den_final")}
small.set_option("loop_kernel", 0)
run_solo(8, 4)
out["kernels_us_gflops"]["latency@B64"] = {n: tk(small, n, 64, 196, 200, stream) for n in ("den_qkv", "den_outproj", "den_ffn1", "den_ffn2", "den_final")}
# parity coalesced vs per request
big.sample_many(reqs[:NREQ]); torch.cuda.synchronize()
for i in range(NREQ):
    q = reqs[i]
    small.sample(q["text_emb"], q["init_latents"], q["lengths"], None, None, solo[i])
torch.cuda.synchronize()
out["max_abs_joints_coalesced_vs_per_request"] = max(float((reqs[i]["joints_out"] - solo[i]).abs().max()) for i in range(NREQ))
print(json.dumps(out))
