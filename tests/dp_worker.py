"""Worker for test_dp_gloo.py: one rank of a world_size-N data-parallel sampling job on CPU (gloo).
Each rank: receive the packed weights by ONE broadcast from rank 0, load them into its own engine
(the TEST-ONLY simulator here; libmldhip on GPUs), sample its contiguous shard of the prompts, and
gather the motions on rank 0, which saves them for the parent test to compare with a 1-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [HERE, REPO, os.path.join(REPO, "motion-latent-diffusion_amd")]
import simlib  # noqa: E402
from mld_hip import dp, synthetic as syn  # noqa: E402


def sampler_job(mode, out_path, n):
    """mode 'action' / 'novae': this rank's shard through mld_hip.MLD + DataParallelSampler (the drop-in surface), the
    weights from ONE packed broadcast, starting noise pinned per prompt; rank 0 saves what every rank produced."""
    import dp_models
    mode, _, shard = mode.partition(":")          # "action:pack": the sampler packs the prompts onto as few ranks as hold them
    rank, world = dist.get_rank(), dist.get_world_size()
    template = dp_models.state_template(mode)
    src = template if rank == 0 else {k: np.full_like(v, np.nan) for k, v in template.items()}
    state = dp.broadcast_state(src, template, torch.device("cpu"), src=0)
    model, close = dp_models.build(mode, {k: v.clone() for k, v in state.items()}, key=f"inject:dp_{mode}_{rank}")
    try:
        kw = dp_models.job(mode, n)
        idx, motions = dp.DataParallelSampler(model, batch_size=2, shard=shard or "auto", verbose=True)(**kw)
    finally:
        close()
    gathered = [None] * world
    dist.all_gather_object(gathered, (idx, [np.asarray(m) for m in motions]))
    if rank == 0:
        np.savez(out_path, **{f"m_{i}": m for ids, ms in gathered for i, m in zip(ids, ms)})
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_path, nprompts = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    if len(sys.argv) > 3:
        return sampler_job(sys.argv[3], out_path, nprompts)
    rank, world = dist.get_rank(), dist.get_world_size()
    dims = syn.ModelDims(num_layers=3)            # a 3-layer skip stack: the smallest the engine builds, sized for the CPU suite
    template = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict(dims=dims).items()},
                **{"vae." + k: v for k, v in syn.make_vae_state_dict(dims=dims).items()}}
    mean, std = syn.make_mean_std()
    template["mean"], template["std"] = mean, std
    # only rank 0 holds real values; the others must get them from the broadcast
    src = template if rank == 0 else {k: np.full_like(v, np.nan) for k, v in template.items()}
    state = dp.broadcast_state(src, template, torch.device("cpu"), src=0)
    eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=4, max_frames=16, num_inference_steps=2, num_layers=3)
    eng.load_state_dict(state)
    eng.finalize()
    batch = syn.make_batch(nprompts, [16, 13, 7, 16, 9][:nprompts], seed=77)
    lo, hi = dp.shard_range(nprompts, rank, world)
    B = hi - lo
    T = max(batch.lengths[lo:hi]) if B else 0
    joints = np.zeros((B, T, 22, 3), np.float32)
    if B:
        text = np.concatenate([batch.text_emb[lo:hi], batch.text_emb[nprompts + lo:nprompts + hi]])
        eng.sample(text, batch.init_latents[lo:hi].copy(), batch.lengths[lo:hi], None, None, joints)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, joints))
    if rank == 0:
        np.savez(out_path, **{f"j_{l}_{h}": j for l, h, j in gathered})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
