"""GPU A/B of the decoder's opt-in half-Q|K|V self-attention block ("dec_half", kernels/dec_half.hpp) against the default fp32-Q|K|V x3 form:
  * decode time (mldhip_vae_decode + mldhip_feats2joints on HIP events) at N = 1 280 and 64 motions x 196 frames, interleaved rounds in ONE process,
  * joints against the exact-fp32 engine on the same latents: the committed fixture's latents (|z| ~ 75) and unit-normal latents, on BOTH synthetic
    weight families -- the GPU side of tools/precision_attribution_decoder.py's emulation (profiles/r06_decoder_precision.json),
  * what finalize's probe reads for the form (mldhip_numeric_status) on each family.
Prints one JSON document (-> profiles/r06_decoder_half_ab.json)."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
T = 196
mean, std = syn.make_mean_std()
gold = os.path.join(ROOT, "tests", "golden")
fam1_lat = np.load(os.path.join(gold, "pipeline_b64.npz"))["latents"]            # [64, 1, 256]
out = {"what": __doc__.split("\n")[0], "T": T, "timing": {}, "errors": {}, "probe": {}}


def engine(prec, sdv, nmax, dec_half=0, probe=True):
    e = _lib.Engine(device=0, max_batch=nmax, max_frames=T, precision=prec)
    e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(sdv, "vae.")
    e.load_tensor("mean", mean); e.load_tensor("std", std)
    if dec_half:
        e.set_option("dec_half", dec_half)
    if not probe:
        e.set_option("range_probe", 0)            # timing handles: no verdict, no veto
    e.finalize()
    return e


def decode(e, z, lens, feats, joints):
    e.vae_decode(z, lens, feats)
    e.feats2joints(feats, len(lens), T, joints)


def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(min(ts))


sdv1 = syn.make_vae_state_dict()
sdv2 = syn.trained_like(sdv1)
# ---- timing, family 1 (the arithmetic does not depend on the values)
for N in ((1280,) if os.environ.get("AB_TIMING_ONLY") else (1280, 64)):
    e = engine(1, sdv1, N, probe=False)
    z = torch.from_numpy(np.tile(fam1_lat, (N // 64, 1, 1))).to(dev)
    lens = [T] * N
    feats = torch.empty(N, T, 263, device=dev); joints = torch.empty(N, T, 22, 3, device=dev)
    rows = {}
    for rnd in range(3):
        for name, v in (("x3_fp32_qkv", 0), ("half_qkv_64row", 2), ("half_qkv_96row", 6)):
            e.set_option("dec_half", v)
            med, best = timed(lambda: decode(e, z, lens, feats, joints))
            rows.setdefault(name, []).append(round(med, 3))
    out["timing"][f"{N}_motions"] = {k: {"ms_median_per_round": v, "ms": min(v)} for k, v in rows.items()}
    print(N, out["timing"][f"{N}_motions"], flush=True)
    e.close()

if os.environ.get("AB_TIMING_ONLY"):
    print(json.dumps(out)); sys.exit(0)
# ---- errors against the exact-fp32 engine, and the probe's reading
B = 64
g = syn._rng(31, "ab_unit").standard_normal((B, 1, 256)).astype(np.float32)
lens = ([196, 196, 120, 64, 196, 33, 196, 150] * 8)[:B]
for fam, sdv in (("family1", sdv1), ("family2_trained_like", sdv2)):
    ref = engine(0, sdv, B)
    x3 = engine(1, sdv, B)
    hf = engine(1, sdv, B, dec_half=1)
    out["probe"][fam] = {k: v for k, v in hf.numeric_status().items()}
    hf.set_option("dec_half", 2)
    for zname, zz in (("fixture_latents", fam1_lat), ("unit_latents", g)):
        z = torch.from_numpy(zz).to(dev)
        res = {}
        for name, e in (("f32", ref), ("x3", x3), ("half", hf)):
            feats = torch.zeros(B, T, 263, device=dev); joints = torch.zeros(B, T, 22, 3, device=dev)
            decode(e, z, lens, feats, joints); torch.cuda.synchronize()
            res[name] = (feats.cpu().numpy(), joints.cpu().numpy())
        row = {}
        for name in ("x3", "half"):
            ef = max(float(np.abs(res[name][0][i, :n] - res["f32"][0][i, :n]).max()) for i, n in enumerate(lens))
            ej = max(float(np.abs(res[name][1][i, :n] - res["f32"][1][i, :n]).max()) for i, n in enumerate(lens))
            row[name] = {"feats_max_abs_vs_f32_engine": ef, "joints_max_abs_vs_f32_engine": ej}
        row["feats_absmax"] = float(np.abs(res["f32"][0]).max())
        out["errors"][f"{fam}/{zname}"] = row
        print(fam, zname, row, flush=True)
    for e in (ref, x3, hf):
        e.close()
print(json.dumps(out))
