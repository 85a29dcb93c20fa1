"""GPU A/B of two BUILDS of the engine in one process: mld_hip/libmldhip.so against mld_hip/libmldhip_alt.so (the same sources compiled
with extra flags:  make -C motion-latent-diffusion_amd/csrc alt HIPCC_EXTRA=-fno-slp-vectorize).  Headline call shape (N motions per
mldhip_sample_many call, split-f16), interleaved rounds, best of 4 per round: whole call, loop-only call, joints of request 0 against
the reference fixture and the two builds against each other.  Prints one JSON line.

Why -fno-slp-vectorize is the first candidate: hipcc's SLP vectoriser packs the paired scalar f32 operations of the GELU / LayerNorm
epilogues into v_pk_add / v_pk_mul / v_pk_fma_f32 (tools/isa_report.py: 548 of the persistent loop's 3 831 VALU instructions, 494 of the
decoder tail's 2 213), and MI355X_MICROARCH.md prices a packed f32 instruction beside matrix instructions at +22 .. +26 cycles over
its two scalar halves -- in exactly the phases (the loop's feed-forward stage, the tail's GELU between linear2's items) that run at
56 % / 49 % of the matrix rate."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
N = int(os.environ.get("AB_N", "2048"))
ALT = os.path.join(os.path.dirname(_lib.DEFAULT_LIB), "libmldhip_alt.so")
assert os.path.exists(ALT), "build it first: make -C motion-latent-diffusion_amd/csrc alt HIPCC_EXTRA=..."
engines = {}
for name, path in (("default", None), ("alt", ALT)):
    e = _lib.Engine(lib=_lib.load_library(path), device=0, max_batch=N, max_frames=196, precision=1)
    e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    engines[name] = e
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


out, joints = {"rounds": []}, {}
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    row = {}
    for name, e in engines.items():
        row[name + "_loop_ms"] = round(best(lambda: e.sample_many(lat_only)) * 1e3, 3)
        row[name + "_all_ms"] = round(best(lambda: e.sample_many(reqs)) * 1e3, 3)
        joints[name] = torch.cat([q["joints_out"] for q in reqs]).clone()
        row[name + "_joints_err"] = float(np.abs(reqs[0]["joints_out"].cpu().numpy()[:, ::4] - g["joints_every4"]).max())
    out["rounds"].append(row)
    print(row, flush=True)
out["builds_max_abs_diff"] = float((joints["default"] - joints["alt"]).abs().max().item())
print(json.dumps(out))
