import sys, os, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn
steps = int(os.environ.get("NOVAE_STEPS", "20"))
for nfl in (1, 2):
    eng = _lib.Engine(device=0, max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                      scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0, max_in_flight=nfl)
    eng.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    m, s = syn.make_mean_std(); eng.load_tensor("mean", m); eng.load_tensor("std", s); eng.finalize()
    b = syn.make_batch(64); dev = torch.device("cuda:0")
    text = torch.from_numpy(b.text_emb).to(dev)
    xs = [torch.randn(64, 196, 263, device=dev) for _ in range(nfl)]; js = [torch.empty(64, 196, 22, 3, device=dev) for _ in range(nfl)]
    sts = [torch.cuda.Stream() for _ in range(nfl)]
    torch.cuda.synchronize()
    for i in range(nfl): eng.sample_novae(text, xs[i], b.lengths, None, 1, None, js[i], sts[i].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2 * nfl): eng.sample_novae(text, xs[i % nfl], b.lengths, None, 1, None, js[i % nfl], sts[i % nfl].cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"in_flight": nfl, "ms_per_ddpm_step_per_batch": dt * 1e3 / (2 * nfl * steps)}))
    eng.close()
