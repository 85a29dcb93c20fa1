"""Full-length fixture for BASELINE config 4 (config_novae_humanml3d.yaml: raw-motion diffusion, 1000 DDPM steps).

Run (build container only, needs /root/reference; CPU; roughly an hour):
    python oracle/make_golden_novae1000.py ref      # the reference's own MldDenoiser in the loop of mld.py:290-360, fp32
    python oracle/make_golden_novae1000.py f64      # the oracle restatement in float64 (the fp32 noise floor's yardstick)
    python oracle/make_golden_novae1000.py merge    # -> tests/golden/novae_pipeline_1000.npz

B = 2 motions, lengths [196, 150], CFG 7.5, 1000 DDPM steps (configs/modules_novae/scheduler.yaml:16-29).  The per-step
Gaussian draws are NOT stored (1000 x 2 x 196 x 263 floats = 412 MB): they are the engine's counter-based stream
``philox_normal(seed, step)`` (oracle.mld_oracle.philox_normal restates kernels/novae.hpp), regenerated here and inside the
engine from (seed, step index).  The fixture keeps the final features and joints of the reference run, the float64 oracle
run's distance from it (= how far two correct fp32/fp64 evaluations of this chaotic 1000-step map drift apart), also
measured after 10 / 100 / 500 steps.

Scheduler: oracle.mld_oracle.DDPMSchedule (diffusers absent -> restated, PARITY UNPINNED; pinned against Ho et al.'s
closed forms in tests/test_scheduler_identities.py)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle._paths  # noqa: E402,F401
from mld_hip import synthetic as syn  # noqa: E402
from oracle import mld_oracle as O  # noqa: E402

REF = os.environ.get("MLD_REFERENCE", "/root/reference")
OUT = os.path.join(oracle._paths.REPO, "tests", "golden")
TMP = os.environ.get("MLD_GOLDEN_TMP", "/tmp/novae1000")
B, T, NF, STEPS, SEED, GUIDANCE = 2, 196, 263, 1000, 20240924, 7.5
LENGTHS = [196, 150]
SNAPS = (10, 100, 500)


def inputs():
    b = syn.make_batch(B, LENGTHS, seed=41)
    lat0 = syn._rng(42, "nv1000").standard_normal((B, T, NF)).astype(np.float32)
    return b, lat0


def noise(i):
    return O.philox_normal(B * T * NF, SEED, i).reshape(B, T, NF)


@torch.no_grad()
def run_ref():
    sys.path.insert(0, REF)
    from mld.models.architectures.mld_denoiser import MldDenoiser
    from mld.data.humanml.scripts.motion_process import recover_from_ric

    class Abl:
        SKIP_CONNECT = True
        VAE_TYPE = "no"
        PE_TYPE = "mld"
        DIFF_PE_TYPE = "mld"
        MLP_DIST = False

    den = MldDenoiser(ablation=Abl, nfeats=NF, condition="text", latent_dim=[1, 512], ff_size=1024, num_layers=9, num_heads=4,
                      arch="trans_dec", text_encoded_dim=768).eval()
    den.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_novae_denoiser_state_dict().items()}, strict=True)
    b, lat0 = inputs()
    mean, std = syn.make_mean_std()
    sch = O.DDPMSchedule()
    lat = torch.from_numpy(lat0)
    enc = torch.from_numpy(b.text_emb)
    snaps = {}
    t0 = time.time()
    for i, t in enumerate(sch.set_timesteps(STEPS)):
        eps = den(sample=torch.cat([lat] * 2), timestep=torch.tensor(int(t)), encoder_hidden_states=enc, lengths=LENGTHS * 2)[0]
        u, c = eps.chunk(2)
        lat = torch.from_numpy(np.asarray(sch.step((u + GUIDANCE * (c - u)).numpy(), int(t), lat.numpy(), noise(i)), np.float32))
        if i + 1 in SNAPS:
            snaps[f"lat_after_{i + 1}"] = lat.numpy().copy()
        if i % 50 == 0:
            print(f"ref step {i} {time.time() - t0:.0f}s |x|max {float(lat.abs().max()):.3f}", flush=True)
    feats = lat.numpy()
    joints = recover_from_ric(torch.from_numpy(feats) * torch.from_numpy(std) + torch.from_numpy(mean), 22).numpy()
    np.savez(os.path.join(TMP, "ref.npz"), feats=feats, joints=joints, **snaps)


def run_f64():
    ops = O.TorchOps("float64")
    bd = O.to_backend(ops, syn.make_novae_denoiser_state_dict())
    b, lat0 = inputs()
    sch = O.DDPMSchedule()
    lat = ops.asarray(lat0)
    te = ops.asarray(b.text_emb)
    snaps = {}
    t0 = time.time()
    for i, t in enumerate(sch.set_timesteps(STEPS)):
        eps = O.denoiser_forward_novae(ops, bd, ops.cat([lat, lat], 0), t, te, LENGTHS * 2)
        u, c = eps[:B], eps[B:]
        lat = sch.step(u + GUIDANCE * (c - u), int(t), lat, ops.asarray(noise(i)))
        if i + 1 in SNAPS:
            snaps[f"lat_after_{i + 1}"] = ops.to_numpy(lat).copy()
        if i % 50 == 0:
            print(f"f64 step {i} {time.time() - t0:.0f}s", flush=True)
    mean, std = syn.make_mean_std()
    joints = O.feats2joints(ops, lat, ops.asarray(mean), ops.asarray(std))
    np.savez(os.path.join(TMP, "f64.npz"), feats=ops.to_numpy(lat), joints=ops.to_numpy(joints), **snaps)


def merge():
    r, d = np.load(os.path.join(TMP, "ref.npz")), np.load(os.path.join(TMP, "f64.npz"))
    valid = np.zeros((B, T), bool)
    for i, n in enumerate(LENGTHS):
        valid[i, :n] = True
    out = dict(lengths=np.array(LENGTHS), seed=np.int64(SEED), steps=np.int64(STEPS), batch_seed=np.int64(41), lat0_seed=np.int64(42),
               feats=r["feats"], joints=r["joints"].astype(np.float32),
               f64_diff_feats=np.abs(r["feats"] - d["feats"])[valid].max(), f64_diff_joints=np.abs(r["joints"] - d["joints"])[valid].max(),
               feats_absmax=np.abs(r["feats"]).max())
    for s in SNAPS:          # the snapshots themselves stay in TMP (412 KB each); their fp32-vs-fp64 distances document the drift
        out[f"f64_diff_after_{s}"] = np.abs(r[f"lat_after_{s}"] - d[f"lat_after_{s}"])[valid].max()
    np.savez_compressed(os.path.join(OUT, "novae_pipeline_1000.npz"), **out)
    print({k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(TMP, exist_ok=True)
    torch.set_num_threads(int(os.environ.get("MLD_THREADS", "3")))
    {"ref": run_ref, "f64": run_f64, "merge": merge}[sys.argv[1]]()
