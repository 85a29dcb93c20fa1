"""(GPU, round 6) config 4 (diffusion-only, bs 64, T = 196): the folded cross-attention sub-layer ("cross_fold" 1: LayerNorm 1 + two-token cross-attention + LayerNorm 2 as ONE launch,
kernels/novae.hpp) against the five launches, one engine, interleaved rounds, one and two batches in flight; joints of a 100-step run against each other.  One JSON line."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
B, T, STEPS = 64, 196, int(os.environ.get("AB_STEPS", "100"))
out = {"steps": STEPS, "ms_per_ddpm_step": {}}
for prec, pname in ((1, "f16x3"), (0, "f32")):
    e = _lib.Engine(device=0, max_batch=B, max_frames=T, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC, scheduler_type=_lib.SCHED_DDPM,
                    num_inference_steps=STEPS, steps_offset=0, precision=prec, max_in_flight=2)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    b = syn.make_batch(B, None, seed=1234, max_len=T)
    text = torch.from_numpy(b.text_emb).to(dev)
    x0 = [torch.randn(B, T, 263, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + i)) for i in range(2)]
    joints = [torch.empty(B, T, 22, 3, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def run(nfl):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(nfl):
            e.sample_novae(text, x0[i], b.lengths, None, 99 + i, None, joints[i], streams[i].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / (STEPS * nfl)
    res, keep = {}, {}
    for rnd in range(3):
        for cf in (1, 0):
            e.set_option("cross_fold", cf)
            run(2)                                   # (graphs of this option captured)
            for nfl in (1, 2):
                res.setdefault((cf, nfl), []).append(run(nfl))
            keep[cf] = joints[0].clone()
            if rnd == 0:                             # one denoiser call (no chaotic 100-step map in between): the algebra itself
                xs = torch.randn(2 * B, T, 263, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
                eps = torch.empty(2 * B, T, 263, device=dev)
                e.denoiser_forward_novae(xs, 500, text, b.lengths * 2, T, eps); torch.cuda.synchronize()
                keep[("eps", cf)] = eps.clone()
    out["ms_per_ddpm_step"][pname] = {"cross_fold_%d_in_flight_%d" % k: round(min(v), 3) for k, v in res.items()}
    out.setdefault("joints_max_abs_fold_vs_unfolded_%d_steps" % STEPS, {})[pname] = float((keep[1] - keep[0]).abs().max())
    out.setdefault("one_denoiser_call_max_abs_fold_vs_unfolded", {})[pname] = [float((keep[("eps", 1)] - keep[("eps", 0)]).abs().max()), float(keep[("eps", 0)].abs().max())]
    print(pname, out["ms_per_ddpm_step"][pname], out["one_denoiser_call_max_abs_fold_vs_unfolded"][pname], flush=True)
    e.close()
print(json.dumps(out))
