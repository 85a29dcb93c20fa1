"""A/B of the diffusion-only variant's kernels on ONE box, split-f16 mode, BASELINE config 4 shape (bs 64, T = 196, d = 512), one and two
batches in flight, interleaved rounds: option "gemm_pipe" 1 (software-pipelined 128 x 256 GEMM tile, kernels/gemm_pipe.hpp) against 0 (the
64 x 128 staged tile of kernels/gemm.hpp) -- same products in the same order: the joints must be IDENTICAL -- and "flash_attn" 1 (key-blocked
head-dim-128 attention, attention.hpp attn_flash128_x3_kernel) against 0 (the two-phase kernel of novae.hpp): another summation order, the
difference is reported.    python tools/ab_novae_gemm.py  [NOVAE_STEPS=40]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn

steps = int(os.environ.get("NOVAE_STEPS", "40"))
dev = torch.device("cuda:0")
b = syn.make_batch(64)
text = torch.from_numpy(b.text_emb).to(dev)
g = torch.Generator(device="cpu").manual_seed(5)
x0 = [torch.randn(64, 196, 263, generator=g).to(dev) for _ in range(2)]
VARIANTS = {"pipe1_flash1": {"gemm_pipe": 1, "flash_attn": 1}, "pipe1_flash0": {"gemm_pipe": 1, "flash_attn": 0}, "pipe0_flash0": {"gemm_pipe": 0, "flash_attn": 0}}
engs = {}
for pipe, opts in VARIANTS.items():
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0, max_in_flight=2, precision=1)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s)
    for k, v in opts.items(): e.set_option(k, v)
    e.finalize()
    engs[pipe] = e
sts = [torch.cuda.Stream() for _ in range(2)]
js = {p: [torch.empty(64, 196, 22, 3, device=dev) for _ in range(2)] for p in engs}
res = {"steps": steps, "rounds": []}
for p, e in engs.items():                                   # warm-up (graph capture) + the outputs that are compared
    for i in range(2): e.sample_novae(text, x0[i], b.lengths, None, 1, None, js[p][i], sts[i].cuda_stream)
torch.cuda.synchronize()
res["max_abs_joint_difference_pipe_vs_staged_tile"] = float((js["pipe1_flash0"][0] - js["pipe0_flash0"][0]).abs().max())
res["max_abs_joint_difference_key_blocked_vs_two_phase_attention"] = float((js["pipe1_flash1"][0] - js["pipe1_flash0"][0]).abs().max())
res["joints_absmax"] = float(js["pipe1_flash1"][0].abs().max())
res["joints_finite"] = bool(torch.isfinite(js["pipe1_flash1"][0]).all())
for rnd in range(3):
    row = {}
    for nfl in (1, 2):
        for p, e in engs.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(2 * nfl): e.sample_novae(text, x0[i % nfl], b.lengths, None, 1, None, js[p][i % nfl], sts[i % nfl].cuda_stream)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            row[f"{p}_in_flight{nfl}_ms_per_ddpm_step"] = round(dt * 1e3 / (2 * nfl * steps), 4)
    res["rounds"].append(row)
best = {k: min(r[k] for r in res["rounds"]) for k in res["rounds"][0]}
res["best"] = best
res["tflops_at_best"] = {k: round(1.29e3 / v, 1) for k, v in best.items()}      # 1.29 TFLOP per DDPM step
print(json.dumps(res))
for e in engs.values(): e.close()
