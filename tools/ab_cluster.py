"""The reverse loop of ONE request on the launch family (tile32.hpp, loop_kernel 1), the sample-major persistent loop (3) and the cluster
loop (loop_cluster.hpp, loop_kernel 4; write-through and plain payload stores) on one box.
  python tools/ab_cluster.py [--batches 64,128,8] [--rounds 5] [--out gpurun_out/r05_cluster_ab.json]
Reported per batch size and variant: ms per loop-only call (latents out only) and per full call (decode + joints), median / min over interleaved rounds,
latents max-abs against the exact-fp32 engine and (batch 64 only, when --oracle) against the CPU oracle; the numeric status after the runs."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np
import torch
from mld_hip import _lib, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_cluster_ab.json"))
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batches", default="64,128,8")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--oracle", type=int, default=0)
    ap.add_argument("--variants", default="")
    ap.add_argument("--nolat", type=int, default=0)
    ap.add_argument("--skipref", type=int, default=0, help="debug: leave the exact-fp32 engine out of the timing rounds")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    mean, std = syn.make_mean_std()
    out = {"what": __doc__.split("\n")[0], "rounds": a.rounds, "steps": a.steps, "batches": {}}
    for B in [int(x) for x in a.batches.split(",")]:
        bb = syn.make_batch(B, [196] * B)
        te, x0 = torch.from_numpy(bb.text_emb).to(dev), torch.from_numpy(bb.init_latents).to(dev)
        variants = [("f32_launches", 0, {"loop_kernel": 1}), ("x3_launches", 1, {"loop_kernel": 1}), ("x3_persistent", 1, {"loop_kernel": 3}),
                    ("x3_cluster_wt", 1, {"loop_kernel": 4, "cluster_wt": 1}), ("x3_cluster_plain", 1, {"loop_kernel": 4, "cluster_wt": 0}),
                    ("x3_cluster_4groups", 1, {"loop_kernel": 4, "cluster_wt": 0, "cluster_groups": 4}),
                    ("x3_cluster_tail48", 1, {"loop_kernel": 4, "cluster_wt": 0, "ffn_strip": 3})]
        if a.variants:
            variants = [v for v in variants if v[0] in a.variants.split(",") or v[0] == "f32_launches"]
        engines = {}
        for name, prec, opts in variants:
            e = _lib.Engine(device=0, max_batch=B, max_frames=196, precision=prec, num_inference_steps=a.steps)
            e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std); e.finalize()
            for k, v in opts.items():
                e.set_option(k, v)
            lat, j = torch.empty(B, 1, 256, device=dev), torch.empty(B, 196, 22, 3, device=dev)
            try:
                e.sample(te, x0, bb.lengths, lat); torch.cuda.synchronize()
                l_first = lat.cpu().numpy().copy()
                e.sample(te, x0, bb.lengths, lat, None, j); torch.cuda.synchronize()
                if "f32_launches" in engines:
                    lr0 = engines["f32_launches"][1].cpu().numpy()
                    print(B, name, "first call (latents only) vs f32 engine %.3e, second (full) %.3e" % (np.abs(l_first - lr0).max(), np.abs(lat.cpu().numpy() - lr0).max()), flush=True)
            except Exception as ex:       # noqa: BLE001
                print(B, name, "FAILED", ex, flush=True)
                continue
            engines[name] = (e, lat, j, [], [])
        for _ in range(a.rounds):
            for name, (e, lat, j, tl, tf) in engines.items():
                if a.skipref and name == "f32_launches":
                    tl.append(0.0); tf.append(0.0)
                    continue
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(0 if a.nolat else 4):
                    e.sample(te, x0, bb.lengths, lat)
                torch.cuda.synchronize(); tl.append((time.perf_counter() - t0) / 4 * 1e3)
                if os.environ.get("AB_DEBUG"):
                    print(name, "after 4 latent-only calls: vs f32 %.3e" % np.abs(lat.cpu().numpy() - engines["f32_launches"][1].cpu().numpy()).max(), flush=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k4 in range(4):
                    e.sample(te, x0, bb.lengths, lat, None, j)
                    if os.environ.get("AB_DEBUG") == "2":
                        torch.cuda.synchronize()
                        print(name, "full call", k4, "vs f32 %.3e" % np.abs(lat.cpu().numpy() - engines["f32_launches"][1].cpu().numpy()).max(), flush=True)
                torch.cuda.synchronize(); tf.append((time.perf_counter() - t0) / 4 * 1e3)
                if os.environ.get("AB_DEBUG"):
                    print(name, "after 4 full calls: vs f32 %.3e" % np.abs(lat.cpu().numpy() - engines["f32_launches"][1].cpu().numpy()).max(), flush=True)
        lref = engines["f32_launches"][1].cpu().numpy()
        jref = engines["f32_launches"][2].cpu().numpy()
        lor = None
        if a.oracle and B <= 64:
            from oracle import mld_oracle as O
            ops = O.TorchOps("float32")
            lor = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, sdd), ops.asarray(bb.text_emb), ops.asarray(bb.init_latents), 7.5, a.steps, 4))
        res = {}
        for name, (e, lat, j, tl, tf) in engines.items():
            ln = lat.cpu().numpy()
            res[name] = {"loop_ms_median": round(float(np.median(tl)), 4), "loop_ms_min": round(min(tl), 4), "full_ms_median": round(float(np.median(tf)), 4),
                         "full_ms_min": round(min(tf), 4), "motions_per_s_full": round(B / (float(np.median(tf)) * 1e-3), 1),
                         "latents_max_abs_vs_f32_engine": float(np.abs(ln - lref).max()), "joints_max_abs_vs_f32_engine": float(np.abs(j.cpu().numpy() - jref).max()),
                         "nonfinite": int(np.isnan(ln).sum()), "launches": list(e.launch_counts()), "numeric": e.numeric_status() if hasattr(e, "numeric_status") else None}
            if lor is not None:
                res[name]["latents_max_abs_vs_oracle"] = float(np.abs(ln - lor).max())
            print(B, name, res[name], flush=True)
        out["batches"][str(B)] = res
        for name, (e, *_rest) in engines.items():
            e.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1, default=str)


if __name__ == "__main__":
    main()
