"""Worker for test_gpu_parity.py::test_nccl_broadcast_and_dp_sampler: one rank of a data-parallel job on real GPUs,
RCCL backend ("nccl" on ROCm).  Rank 0 holds the weights; every other rank starts from NaNs and must receive them
through the ONE packed broadcast of mld_hip.dp.broadcast_state; every rank then samples its shard of the prompts on
its own GPU through mld_hip.MLD + DataParallelSampler and rank 0 saves the gathered motions."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [REPO, os.path.join(REPO, "motion-latent-diffusion_amd")]
from mld_hip import dp, synthetic as syn  # noqa: E402
from mld_hip import engine as E  # noqa: E402


def main():
    out_path, nprompts = sys.argv[1], int(sys.argv[2])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    template = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()},
                **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
    src = template if rank == 0 else {k: np.full_like(v, np.nan) for k, v in template.items()}
    state = dp.broadcast_state(src, template, dev, src=0)                # RCCL broadcast (executed at world == 1 too)
    assert all(torch.isfinite(v).all().item() for v in state.values()), "broadcast did not deliver the weights"
    # the drop-in model on this rank's GPU, weights from the broadcast blob
    from mld_hip.config import load_config
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder
    E.configure("text", max_batch=4, max_frames=64, max_in_flight=2)
    cfg = load_config(os.path.join(REPO, "motion-latent-diffusion_amd", "configs", "config_mld_humanml3d.yaml"))
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev)
    sd = {k: v for k, v in state.items() if k.startswith(("denoiser.", "vae."))}
    model.load_state_dict(sd, strict=False)
    texts = ["prompt %d" % i for i in range(nprompts)]
    lengths = [24 + 8 * (i % 5) for i in range(nprompts)]
    # starting noise pinned PER PROMPT (the same tensor on every rank): a motion then does not depend on the rank / chunk it lands in,
    # and the parent test compares every rank's motions with a single-process run of the same prompts
    lat0 = torch.from_numpy(syn._rng(4242, "nccl_dp").standard_normal((nprompts, 1, 256)).astype(np.float32))
    idx, motions = dp.DataParallelSampler(model, batch_size=4, in_flight=2)(texts, lengths, init_latents=lat0)
    ok = all(m.shape == (lengths[i], 22, 3) and bool(torch.isfinite(m).all()) for i, m in zip(idx, motions))
    ident = (rank, local, str(getattr(torch.cuda.get_device_properties(local), "uuid", local)))
    gathered = [None] * world
    dist.all_gather_object(gathered, (ident, idx, ok, [m.numpy() for m in motions]))
    if rank == 0:
        import json
        json.dump({"world": world, "ranks": [g[0] for g in gathered], "indices": [g[1] for g in gathered],
                   "ok": [g[2] for g in gathered], "backend": dist.get_backend()}, open(out_path, "w"))
        np.savez(out_path + ".npz", **{f"m_{i}": m for g in gathered for i, m in zip(g[1], g[3])})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
