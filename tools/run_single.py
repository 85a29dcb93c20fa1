"""N bs-64 batches, one at a time, on the default split-f16 engine: the process tools/gpu_single_prof.sh wraps in rocprofv3.
  python tools/run_single.py [n] [opt:val ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import torch
from mld_hip import _lib, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1)
e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
mean, std = syn.make_mean_std()
e.load_tensor("mean", mean); e.load_tensor("std", std); e.finalize()
for kv in sys.argv[2:]:
    k, v = kv.split(":"); e.set_option(k, int(v))
bb = syn.make_batch(64)
te, x0 = torch.from_numpy(bb.text_emb).to(dev), torch.from_numpy(bb.init_latents).to(dev)
lat, j = torch.empty(64, 1, 256, device=dev), torch.empty(64, 196, 22, 3, device=dev)
for _ in range(n):
    e.sample(te, x0, bb.lengths, lat, None, j)
torch.cuda.synchronize()
print("done", e.launch_counts())
