"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Run:  python oracle/make_golden.py            (needs /root/reference; CPU; ~1 min)

The reference is Python and cannot travel to the GPU box, so its outputs on the seeded
synthetic weights/inputs of ``mld_hip.synthetic`` are frozen here as small fixtures.
What comes from where:

  MldDenoiser / MldVae / recover_from_ric   imported from /root/reference (the real code)
  DDIM scheduler                            oracle.mld_oracle.DDIMSchedule (diffusers is not
                                            installed -> restated, PARITY UNPINNED)
  orchestration (CFG batching, loop)        the 20 lines below, following mld.py:216-265,290-360

Each fixture also records the oracle-vs-reference max-abs difference measured at
generation time (``oracle_diff_*``), so the pinning evidence is in the file itself.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle._paths  # noqa: E402,F401
from mld_hip import synthetic as syn  # noqa: E402
from oracle import mld_oracle as O  # noqa: E402

REF = os.environ.get("MLD_REFERENCE", "/root/reference")
OUT = os.path.join(oracle._paths.REPO, "tests", "golden")


def build_reference():
    sys.path.insert(0, REF)
    from mld.models.architectures.mld_denoiser import MldDenoiser
    from mld.models.architectures.mld_vae import MldVae
    from mld.data.humanml.scripts.motion_process import recover_from_ric

    class Abl:  # TRAIN.ABLATION of config_mld_humanml3d.yaml:33-36 (+ base.yaml defaults)
        SKIP_CONNECT = True
        VAE_TYPE = "mld"
        PE_TYPE = "mld"
        DIFF_PE_TYPE = "mld"
        MLP_DIST = False

    den = MldDenoiser(ablation=Abl, nfeats=263, condition="text", latent_dim=[1, 256], ff_size=1024,
                      num_layers=9, num_heads=4, dropout=0.1, normalize_before=False, activation="gelu",
                      flip_sin_to_cos=True, position_embedding="learned", arch="trans_enc", freq_shift=0,
                      guidance_scale=7.5, guidance_uncondp=0.1, text_encoded_dim=768).eval()
    vae = MldVae(ablation=Abl, nfeats=263, latent_dim=[1, 256], ff_size=1024, num_layers=9, num_heads=4,
                 dropout=0.1, arch="encoder_decoder", normalize_before=False, activation="gelu",
                 position_embedding="learned").eval()
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    den.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=True)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()}, strict=True)
    return den, vae, recover_from_ric, sdd, sdv


@torch.no_grad()
def reference_sample(den, vae, recover_from_ric, text_emb, init_latents, lengths, mean, std,
                     guidance=7.5, steps=50):
    """mld.py:290-360 + 237-240 + 264 with the reference modules and the restated DDIM."""
    sch = O.DDIMSchedule()
    lat = torch.from_numpy(init_latents) * sch.init_noise_sigma
    enc = torch.from_numpy(text_emb)
    for t in sch.set_timesteps(steps):
        x = torch.cat([lat] * 2)
        eps = den(sample=x, timestep=torch.tensor(int(t)), encoder_hidden_states=enc, lengths=list(lengths) * 2)[0]
        u, c = eps.chunk(2)
        eps = u + guidance * (c - u)
        sa, sb, pa, pb = (float(v) for v in sch.coeffs(t))
        x0 = (lat - sb * eps) / sa
        lat = pa * x0 + pb * eps
    z = lat.permute(1, 0, 2)
    feats = vae.decode(z, list(lengths))
    joints = recover_from_ric(feats * torch.from_numpy(std) + torch.from_numpy(mean), 22)
    return lat.numpy(), feats.numpy(), joints.numpy()


def main():
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    den, vae, recover_from_ric, sdd, sdv = build_reference()
    ops = O.NumpyOps(np.float32)
    bd, bv = O.to_backend(ops, sdd), O.to_backend(ops, sdv)
    mean, std = syn.make_mean_std()

    # ---- 1. single denoiser calls (first and last DDIM timesteps), B=3 CFG batch
    b3 = syn.make_batch(3, [50, 100, 100])                       # demo/example.txt lengths
    x = np.concatenate([b3.init_latents] * 2)
    den_out = {}
    for t in (981, 1):
        with torch.no_grad():
            ref = den(sample=torch.from_numpy(x), timestep=torch.tensor(t),
                      encoder_hidden_states=torch.from_numpy(b3.text_emb), lengths=b3.lengths * 2)[0].numpy()
        mine = O.denoiser_forward(ops, bd, x, t, b3.text_emb)
        den_out[f"out_t{t}"] = ref
        den_out[f"oracle_diff_t{t}"] = np.abs(ref - mine).max()
    np.savez_compressed(os.path.join(OUT, "denoiser_b3.npz"), sample=x, text_emb=b3.text_emb, **den_out)

    # ---- 2. VAE decode + joints, ragged B=3
    z = syn._rng(7, "golden_z").standard_normal((3, 1, 256)).astype(np.float32)
    with torch.no_grad():
        feats = vae.decode(torch.from_numpy(z).permute(1, 0, 2), b3.lengths).numpy()
        joints = recover_from_ric(torch.from_numpy(feats) * torch.from_numpy(std) + torch.from_numpy(mean), 22).numpy()
    fm = O.vae_decode(ops, bv, z, b3.lengths)
    jm = O.feats2joints(ops, feats, mean, std)
    np.savez_compressed(os.path.join(OUT, "vae_decode_b3.npz"), z=z, lengths=np.array(b3.lengths), feats=feats,
                        joints=joints, oracle_diff_feats=np.abs(feats - fm).max(),
                        oracle_diff_joints=np.abs(joints - jm).max())

    # ---- 3. full pipeline, config 1 shape (B=3, lengths 50/100/100)
    lat, feats, joints = reference_sample(den, vae, recover_from_ric, b3.text_emb, b3.init_latents, b3.lengths, mean, std)
    jo, fo, lo = O.sample(ops, bd, bv, b3.text_emb, b3.init_latents, b3.lengths, mean, std, return_intermediates=True)
    np.savez_compressed(os.path.join(OUT, "pipeline_b3.npz"), text_emb=b3.text_emb, init_latents=b3.init_latents,
                        lengths=np.array(b3.lengths), latents=lat, feats=feats, joints=joints,
                        oracle_diff_latents=np.abs(lat - lo).max(), oracle_diff_feats=np.abs(feats - fo).max(),
                        oracle_diff_joints=np.abs(joints - jo).max())
    print("pipeline_b3 oracle-vs-reference:", np.abs(lat - lo).max(), np.abs(feats - fo).max(), np.abs(joints - jo).max())

    # ---- 4. full pipeline, config 2 shape (B=64, T=196): latents, EVERY frame of the joints (round 5; `joints_every4` stays for the tests written against it),
    #         the last frame of the features
    b64 = syn.make_batch(64)
    lat, feats, joints = reference_sample(den, vae, recover_from_ric, b64.text_emb, b64.init_latents, b64.lengths, mean, std)
    jo, fo, lo = O.sample(ops, bd, bv, b64.text_emb, b64.init_latents, b64.lengths, mean, std, return_intermediates=True)
    np.savez_compressed(os.path.join(OUT, "pipeline_b64.npz"), latents=lat, joints_every4=joints[:, ::4], joints=joints,
                        feats_frame_last=feats[:, -1], oracle_diff_latents=np.abs(lat - lo).max(),
                        oracle_diff_feats=np.abs(feats - fo).max(), oracle_diff_joints=np.abs(joints - jo).max())
    print("pipeline_b64 oracle-vs-reference:", np.abs(lat - lo).max(), np.abs(feats - fo).max(), np.abs(joints - jo).max())

    # ---- 5. scheduler table (self-consistency only: diffusers absent => unpinned)
    sch = O.DDIMSchedule()
    ts = sch.set_timesteps(50)
    np.savez_compressed(os.path.join(OUT, "ddim_table.npz"), timesteps=ts, alphas_cumprod=sch.alphas_cumprod,
                        coeffs=np.array([sch.coeffs(t) for t in ts], np.float32))
    # ---- 5b. VAE encode (scope row 8f.1): mu / std of the reference's Normal, ragged B=3
    fe = syn._rng(9, "golden_feats").standard_normal((3, 100, 263)).astype(np.float32)
    for i, n in enumerate(b3.lengths):
        fe[i, n:] = 0
    with torch.no_grad():
        _, dist = vae.encode(torch.from_numpy(fe), b3.lengths)
    mu_r, std_r = dist.loc.permute(1, 0, 2).numpy(), dist.scale.permute(1, 0, 2).numpy()
    _, mu_o, lv_o = O.vae_encode(ops, bv, fe, b3.lengths)
    np.savez_compressed(os.path.join(OUT, "vae_encode_b3.npz"), feats=fe, lengths=np.array(b3.lengths), mu=mu_r, std=std_r,
                        oracle_diff_mu=np.abs(mu_r - mu_o).max(), oracle_diff_std=np.abs(std_r - np.sqrt(np.exp(lv_o))).max())
    print("vae_encode_b3 oracle-vs-reference:", np.abs(mu_r - mu_o).max(), np.abs(std_r - np.sqrt(np.exp(lv_o))).max())

    # ---- 6. checkpoint key contract: names + shapes of the reference modules' state_dicts
    import json
    keys = {"denoiser": {k: list(v.shape) for k, v in den.state_dict().items()},
            "vae": {k: list(v.shape) for k, v in vae.state_dict().items()}}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


@torch.no_grad()
def main_action():
    """Fixtures for the action-conditioned variant (BASELINE config 5, config_mld_humanact12.yaml + modules_humanact12):
    MldDenoiser(condition='action', num_layers=15, nclasses=12) and ActorVae(num_layers=6, nfeats=150) imported from
    the reference; a2m_eval's conditioning (mld.py:716-726) + the restated DDIM orchestrated below."""
    import json
    sys.path.insert(0, REF)
    from mld.models.architectures.actor_vae import ActorVae
    from mld.models.architectures.mld_denoiser import MldDenoiser

    class Abl:
        SKIP_CONNECT = True
        VAE_TYPE = "actor"
        PE_TYPE = "mld"
        DIFF_PE_TYPE = "mld"
        MLP_DIST = False

    dims = syn.ModelDims(num_layers=15, nfeats=150)
    den = MldDenoiser(ablation=Abl, nfeats=150, condition="action", latent_dim=[1, 256], ff_size=1024, num_layers=15,
                      num_heads=4, nclasses=12, guidance_scale=7.5).eval()
    vae = ActorVae(ablation=Abl, nfeats=150, latent_dim=[1, 256], ff_size=1024, num_layers=6, num_heads=4).eval()
    sdd = syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12)
    sdv = syn.make_actor_vae_state_dict()
    den.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=True)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()}, strict=True)
    ops = O.NumpyOps(np.float32)
    bd, bv = O.to_backend(ops, sdd), O.to_backend(ops, sdv)

    # single ops, B=4 (denoiser CFG batch R=8; ragged decode)
    acts, lat0, lens = syn.make_action_batch(4, nframes=60)
    cond = np.concatenate([np.zeros_like(acts), acts])
    x = np.concatenate([lat0, lat0])
    ref = den(sample=torch.from_numpy(x), timestep=torch.tensor(981),
              encoder_hidden_states=torch.from_numpy(cond[:, None].astype(np.float32)), lengths=lens * 2)[0].numpy()
    mine = O.denoiser_forward_action(ops, bd, x, 981, cond)
    z = syn._rng(5, "az").standard_normal((4, 1, 256)).astype(np.float32)
    lens2 = [60, 45, 60, 12]
    fr = vae.decode(torch.from_numpy(z).permute(1, 0, 2), lens2).numpy()
    fm = O.actor_decode(ops, bv, z, lens2)
    np.savez_compressed(os.path.join(OUT, "action_ops_b4.npz"), sample=x, cond=cond.astype(np.int32), out_t981=ref, z=z,
                        lengths=np.array(lens2), feats=fr, oracle_diff_den=np.abs(ref - mine).max(),
                        oracle_diff_feats=np.abs(fr - fm).max())
    print("action ops oracle-vs-reference:", np.abs(ref - mine).max(), np.abs(fr - fm).max())
    # ActorVae.encode: Normal(mu, std) of a ragged batch
    fe = syn._rng(9, "actor_feats").standard_normal((3, 60, 150)).astype(np.float32)
    lens3 = [60, 33, 7]
    for i, n in enumerate(lens3):
        fe[i, n:] = 0
    _, dist = vae.encode(torch.from_numpy(fe), lens3)
    mu_r, std_r = dist.loc.numpy(), dist.scale.numpy()                     # [B, D]
    _, mu_o, lv_o = O.actor_encode(ops, bv, fe, lens3)
    np.savez_compressed(os.path.join(OUT, "actor_encode_b3.npz"), feats=fe, lengths=np.array(lens3), mu=mu_r, std=std_r,
                        oracle_diff_mu=np.abs(mu_r - mu_o[:, 0]).max(), oracle_diff_std=np.abs(std_r - np.sqrt(np.exp(lv_o[:, 0]))).max())
    print("actor_encode_b3 oracle-vs-reference:", np.abs(mu_r - mu_o[:, 0]).max(), np.abs(std_r - np.sqrt(np.exp(lv_o[:, 0]))).max())

    # full pipeline at config 5's shape: B=256, T=60, 50 steps (every 8th sample of feats kept)
    acts, lat0, lens = syn.make_action_batch(256, nframes=60)
    sch = O.DDIMSchedule()
    lat = torch.from_numpy(lat0)
    cond = torch.from_numpy(np.concatenate([np.zeros_like(acts), acts])[:, None].astype(np.float32))
    for t in sch.set_timesteps(50):
        eps = den(sample=torch.cat([lat] * 2), timestep=torch.tensor(int(t)), encoder_hidden_states=cond, lengths=lens * 2)[0]
        u, c = eps.chunk(2)
        eps = u + 7.5 * (c - u)
        sa, sb, pa, pb = (float(v) for v in sch.coeffs(t))
        lat = pa * ((lat - sb * eps) / sa) + pb * eps
    feats = vae.decode(lat.permute(1, 0, 2), lens).numpy()
    fo, lo = O.sample_action(ops, bd, bv, acts, lat0, lens, return_intermediates=True)
    np.savez_compressed(os.path.join(OUT, "action_b256.npz"), latents=lat.numpy(), feats_every8=feats[::8],
                        oracle_diff_latents=np.abs(lat.numpy() - lo).max(), oracle_diff_feats=np.abs(feats - fo).max())
    print("action_b256 oracle-vs-reference:", np.abs(lat.numpy() - lo).max(), np.abs(feats - fo).max())

    kp = os.path.join(OUT, "state_dict_keys.json")
    keys = json.load(open(kp))
    keys["denoiser_action"] = {k: list(v.shape) for k, v in den.state_dict().items()}
    keys["actor_vae"] = {k: list(v.shape) for k, v in vae.state_dict().items()}
    with open(kp, "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)


@torch.no_grad()
def main_novae():
    """Fixtures for the diffusion-only variant (BASELINE config 4, config_novae_humanml3d.yaml + modules_novae):
    MldDenoiser(VAE_TYPE 'no', arch trans_dec, latent_dim [1,512]) imported from the reference; the CFG loop of
    mld.py:290-360 with the restated DDPM (diffusers absent) and injected per-step noise orchestrated below."""
    import json
    sys.path.insert(0, REF)
    from mld.models.architectures.mld_denoiser import MldDenoiser
    from mld.data.humanml.scripts.motion_process import recover_from_ric

    class Abl:
        SKIP_CONNECT = True
        VAE_TYPE = "no"
        PE_TYPE = "mld"
        DIFF_PE_TYPE = "mld"
        MLP_DIST = False

    den = MldDenoiser(ablation=Abl, nfeats=263, condition="text", latent_dim=[1, 512], ff_size=1024, num_layers=9, num_heads=4,
                      arch="trans_dec", text_encoded_dim=768).eval()
    sd = syn.make_novae_denoiser_state_dict()
    den.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    ops = O.NumpyOps(np.float32)
    bd = O.to_backend(ops, sd)
    mean, std = syn.make_mean_std()

    # 1. single denoiser calls: ragged small batch at two timesteps
    g = syn._rng(11, "nv")
    x = g.standard_normal((4, 24, 263)).astype(np.float32)
    te = g.standard_normal((4, 1, 768)).astype(np.float32)
    lens = [24, 17, 24, 9]
    outs = {}
    for t in (999, 0):
        ref = den(sample=torch.from_numpy(x), timestep=torch.tensor(t), encoder_hidden_states=torch.from_numpy(te), lengths=lens)[0].numpy()
        outs[f"out_t{t}"] = ref
        outs[f"oracle_diff_t{t}"] = np.abs(ref - O.denoiser_forward_novae(ops, bd, x, t, te, lens)).max()
    np.savez_compressed(os.path.join(OUT, "novae_denoiser_b4.npz"), sample=x, text_emb=te, lengths=np.array(lens), **outs)
    print("novae denoiser oracle-vs-reference:", outs["oracle_diff_t999"], outs["oracle_diff_t0"])

    # 2. one call at config 4's full CFG shape (R = 128, T = 196): every 16th sample x every 7th frame kept
    b64 = syn.make_batch(64)
    xf = syn._rng(12, "nvfull").standard_normal((64, 196, 263)).astype(np.float32)
    lens64 = [196 - 4 * (i % 8) for i in range(64)]
    ref = den(sample=torch.from_numpy(np.concatenate([xf, xf])), timestep=torch.tensor(500),
              encoder_hidden_states=torch.from_numpy(b64.text_emb), lengths=lens64 * 2)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "novae_denoiser_full.npz"), lengths=np.array(lens64), out_t500_sub=ref[::16, ::7])

    # 3. short pipeline: B=3 ragged, 10 DDPM steps (ratio 100), CFG 7.5, injected noise -> feats, joints
    b3 = syn.make_batch(3, [40, 25, 40])
    g = syn._rng(13, "nvpipe")
    lat0 = g.standard_normal((3, 40, 263)).astype(np.float32)
    noise = g.standard_normal((10, 3, 40, 263)).astype(np.float32)
    sch = O.DDPMSchedule()
    lat = torch.from_numpy(lat0)
    enc = torch.from_numpy(b3.text_emb)
    for i, t in enumerate(sch.set_timesteps(10)):
        eps = den(sample=torch.cat([lat] * 2), timestep=torch.tensor(int(t)), encoder_hidden_states=enc, lengths=b3.lengths * 2)[0]
        u, c = eps.chunk(2)
        lat = torch.from_numpy(np.asarray(sch.step((u + 7.5 * (c - u)).numpy(), int(t), lat.numpy(), noise[i]), np.float32))
    feats = lat.numpy()
    joints = recover_from_ric(torch.from_numpy(feats) * torch.from_numpy(std) + torch.from_numpy(mean), 22).numpy()
    jo, fo = O.sample_novae(ops, bd, b3.text_emb, lat0, b3.lengths, noise, mean, std, steps=10)
    np.savez_compressed(os.path.join(OUT, "novae_pipeline_b3.npz"), text_emb=b3.text_emb, init_latents=lat0, step_noise=noise,
                        lengths=np.array(b3.lengths), feats=feats, joints=joints, oracle_diff_feats=np.abs(feats - fo).max(),
                        oracle_diff_joints=np.abs(joints - jo).max())
    print("novae_pipeline_b3 oracle-vs-reference:", np.abs(feats - fo).max(), np.abs(joints - jo).max(), np.abs(feats).max())
    sch.set_timesteps(1000)
    np.savez_compressed(os.path.join(OUT, "ddpm_table.npz"), coeffs=np.array([sch.coeffs(t) for t in range(1000)], np.float32))

    kp = os.path.join(OUT, "state_dict_keys.json")
    keys = json.load(open(kp))
    keys["denoiser_novae"] = {k: list(v.shape) for k, v in den.state_dict().items()}
    with open(kp, "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    if "--action-only" in sys.argv:
        main_action()
    elif "--novae-only" in sys.argv:
        main_novae()
    else:
        main()
        main_action()
        main_novae()
