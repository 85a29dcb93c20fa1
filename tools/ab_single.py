"""One bs-64 batch at a time (mldhip_sample, the latency kernels of tile32.hpp): A/B of per-handle options on one box.
  python tools/ab_single.py [--out profiles/r03_single_batch_ab.json]
Variants: exact-fp32 engine; split-f16 engine with the latency kernels on fp32 MFMAs ("tile_x3" 0) and on split-f16 MFMAs (1).
Reported: median / min ms per batch over interleaved rounds, joints max-abs vs the exact-fp32 engine."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np
import torch
from mld_hip import _lib, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_single_batch_ab.json"))
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--opts", default="", help="extra variants: name=opt:val,opt:val;name2=...")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    mean, std = syn.make_mean_std()
    bb = syn.make_batch(64)
    te, x0 = torch.from_numpy(bb.text_emb).to(dev), torch.from_numpy(bb.init_latents).to(dev)
    variants = [("f32", 0, {}), ("f16x3_tile_fp32", 1, {"tile_x3": 0}), ("f16x3_tile_x3", 1, {"tile_x3": 1})]
    for spec in filter(None, a.opts.split(";")):
        name, kv = spec.split("=")
        variants.append((name, 1, {k: int(v) for k, v in (p.split(":") for p in kv.split(","))}))
    engines = {}
    for name, prec, opts in variants:
        e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=prec)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std); e.finalize()
        for k, v in opts.items():
            e.set_option(k, v)
        lat, j = torch.empty(64, 1, 256, device=dev), torch.empty(64, 196, 22, 3, device=dev)
        e.sample(te, x0, bb.lengths, lat, None, j); torch.cuda.synchronize()
        engines[name] = (e, lat, j, [])
    for _ in range(a.rounds):
        for name, (e, lat, j, ts) in engines.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                e.sample(te, x0, bb.lengths, lat, None, j)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 4 * 1e3)
    jref = engines["f32"][2].cpu().numpy()
    out = {"what": __doc__.split("\n")[0], "rounds": a.rounds, "variants": {}}
    for name, (e, lat, j, ts) in engines.items():
        out["variants"][name] = {"ms_per_batch_median": round(float(np.median(ts)), 4), "min": round(min(ts), 4), "max": round(max(ts), 4),
                                 "joints_max_abs_vs_f32_engine": float(np.abs(j.cpu().numpy() - jref).max()), "launches": list(e.launch_counts())}
        print(name, out["variants"][name], flush=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
