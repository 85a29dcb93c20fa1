#!/usr/bin/env python
"""Phase breakdown of the sample-major persistent loop (kernels/loop_fused.hpp) from its own cycle counters ("fused_dbg" 5: a
stamp behind s_waitcnt 0 at every phase boundary, summed over the 50 steps x 9 layers per wave).  Run on the GPU box; writes
gpurun_out/loop_phase_trace.json.  The stamps serialise what would otherwise overlap across a boundary: shares, not absolute times."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
N = int(os.environ.get("AB_N", "2048"))
eng = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=N, max_frames=196, precision=1)
eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
reqs = []
for i in range(N // 64):
    b = syn.make_batch(64) if i == 0 else syn.make_batch(64, None, seed=1234 + i)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     latents_out=torch.zeros(64, 1, 256, device=dev)))
eng.set_option("loop_kernel", 3)


def best(n=3):
    eng.sample_many(reqs); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.sample_many(reqs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


out = {"motions": N, "loop_ms_plain_build": round(best(), 2)}
ref = reqs[0]["latents_out"].cpu().numpy().copy()
eng.set_option("fused_dbg", 5)
out["loop_ms_traced_build"] = round(best(), 2)
assert np.array_equal(ref, reqs[0]["latents_out"].cpu().numpy()), "the traced build computes the same latents"
tr = eng.profile_trace("den_loop_phases", 64, 196).astype(np.float64).reshape(64, 8, 16)      # [64 workgroups, 8 waves, 16 counters]
names = ["qkv_products", "scores_softmax_attention_output", "out_projection", "residual_norm1", "feed_forward", "skip_linear_epilogue_and_store", "end_of_step",
         "residual_norm2", "layer_output_store_skip_park", "skip_linear_first_half", "parked_rows_reload", "skip_linear_second_half"]
tot = tr[:, :, 15].mean()
out["cycles_per_wave_total"] = int(tot)
out["share"] = {n: round(float(tr[:, :, i].mean() / tot), 4) for i, n in enumerate(names)}
out["ms_at_traced_total"] = {n: round(float(tr[:, :, i].mean() / tot) * out["loop_ms_traced_build"], 2) for i, n in enumerate(names)}
out["wave_spread"] = {n: [int(tr[:, :, i].min()), int(tr[:, :, i].max())] for i, n in enumerate(names)}
eng.set_option("fused_dbg", 0)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "loop_phase_trace.json"), "w"), indent=1)
