"""GPU A/B of the arithmetic modes (mldhip.h MLDHIP_PREC_*): throughput AND measured error vs the reference-generated
fixtures, for BASELINE configs 2 (text, bs 64), 5 (action, bs 256) and 4 (diffusion-only, per DDPM step)."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn

dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = {0: "f32", 1: "f16x3", 2: "bf16", 3: "fp8_denoiser"}
STEPS = int(os.environ.get("AB_STEPS", "12"))
out = {}


def timed(fn, n, nstreams):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    for i in range(max(2, nstreams)):
        fn(i, streams[i % nstreams])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(i, streams[i % nstreams])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---------------- config 2
g = np.load(os.path.join(GOLD, "pipeline_b64.npz"))
b = syn.make_batch(64)
text, lat0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
res = {}
for prec in (0, 1, 2):
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=prec, max_in_flight=4)
    e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
    m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    lat, feats = torch.empty(64, 1, 256, device=dev), torch.empty(64, 196, 263, device=dev)
    joints = [torch.empty(64, 196, 22, 3, device=dev) for _ in range(4)]
    e.sample(text, lat0, b.lengths, lat, feats, joints[0]); torch.cuda.synchronize()
    r = {"max_abs_latents_vs_reference": float(np.abs(lat.cpu().numpy() - g["latents"]).max()),
         "max_abs_feats_vs_reference": float(np.abs(feats.cpu().numpy()[:, -1] - g["feats_frame_last"]).max()),
         "max_abs_joints_vs_reference": float(np.abs(joints[0].cpu().numpy()[:, ::4] - g["joints_every4"]).max())}
    f = lambda i, st: e.sample(text, lat0, b.lengths, None, None, joints[i % 4], st.cuda_stream)
    dt4, dt1 = timed(f, 2 * STEPS, 4), timed(f, STEPS, 1)
    r.update(motions_per_s_4_in_flight=round(64 / dt4, 1), motions_per_s_single=round(64 / dt1, 1), ms_single=round(dt1 * 1e3, 3))
    res[NAMES[prec]] = r
    e.close()
out["config2_text_bs64"] = res

# ---------------- config 5
ga = np.load(os.path.join(GOLD, "action_b256.npz"))
dims = syn.ModelDims(num_layers=15, nfeats=150)
sdd, sdv = syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), syn.make_actor_vae_state_dict()
res = {}
print("action golden keys", list(ga.keys()), file=sys.stderr)
for prec in (0, 1, 2, 3):
    e = _lib.Engine(device=0, max_batch=256, max_frames=60, condition=_lib.COND_ACTION, nclasses=12, vae_arch=_lib.VAE_ACTOR, vae_num_layers=6,
                    num_layers=15, nfeats=150, precision=prec, max_in_flight=2)
    e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.finalize()
    acts, l0, lens = syn.make_action_batch(256, nframes=60, seed=1234)
    if "actions" in ga.files:
        acts, lens = [int(x) for x in ga["actions"]], [int(x) for x in ga["lengths"]]
        l0 = ga["init_latents"] if "init_latents" in ga.files else l0
    x0 = torch.from_numpy(np.ascontiguousarray(l0)).to(dev)
    lat, feats = torch.empty(256, 1, 256, device=dev), [torch.empty(256, 60, 150, device=dev) for _ in range(2)]
    e.sample_action(acts, x0, lens, lat, feats[0]); torch.cuda.synchronize()
    r = {}
    if "latents" in ga.files:
        r["max_abs_latents_vs_reference"] = float(np.abs(lat.cpu().numpy() - ga["latents"]).max())
    for k in ga.files:
        if k.startswith("feats") and ga[k].shape == (256, 60, 150):
            r["max_abs_feats_vs_reference"] = float(np.abs(feats[0].cpu().numpy() - ga[k]).max())
    f = lambda i, st: e.sample_action(acts, x0, lens, None, feats[i % 2], st.cuda_stream)
    dt2, dt1 = timed(f, STEPS, 2), timed(f, max(4, STEPS // 2), 1)
    r.update(motions_per_s_2_in_flight=round(256 / dt2, 1), motions_per_s_single=round(256 / dt1, 1), ms_single=round(dt1 * 1e3, 3))
    res[NAMES[prec]] = r
    e.close()
out["config5_action_bs256"] = res

# ---------------- config 4
gn = np.load(os.path.join(GOLD, "novae_pipeline_b3.npz"))
wn = syn.make_novae_denoiser_state_dict()
m, s = syn.make_mean_std()
res = {}
for prec in (0, 1, 2):
    r = {}
    e = _lib.Engine(device=0, max_batch=3, max_frames=40, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=10, steps_offset=0, precision=prec)
    e.load_state_dict(wn, "denoiser."); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    e.set_option("gemm_small_m", 0)          # M = 240 rows here: force the staged (precision-aware) GEMMs as at full size
    lens = [int(x) for x in gn["lengths"]]
    feats, joints = torch.empty(3, 40, 263, device=dev), torch.empty(3, 40, 22, 3, device=dev)
    e.sample_novae(torch.from_numpy(gn["text_emb"]).to(dev), torch.from_numpy(gn["init_latents"]).to(dev), lens,
                   torch.from_numpy(gn["step_noise"]).to(dev), 0, feats, joints)
    torch.cuda.synchronize()
    r["max_abs_feats_vs_reference_10step_b3"] = float(max(np.abs(feats.cpu().numpy()[i, :n] - gn["feats"][i, :n]).max() for i, n in enumerate(lens)))
    r["feats_absmax"] = float(np.abs(gn["feats"]).max())
    e.close()
    # full length: 1000 DDPM steps, B = 2 (tests/golden/novae_pipeline_1000.npz; in-kernel Philox noise)
    g1 = np.load(os.path.join(GOLD, "novae_pipeline_1000.npz"))
    l1 = [int(x) for x in g1["lengths"]]
    e = _lib.Engine(device=0, max_batch=2, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=1000, steps_offset=0, precision=prec)
    e.load_state_dict(wn, "denoiser."); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    bb = syn.make_batch(2, l1, seed=int(g1["batch_seed"]))
    x1 = syn._rng(int(g1["lat0_seed"]), "nv1000").standard_normal((2, 196, 263)).astype(np.float32)
    f1, j1 = torch.empty(2, 196, 263, device=dev), torch.empty(2, 196, 22, 3, device=dev)
    e.sample_novae(torch.from_numpy(bb.text_emb).to(dev), torch.from_numpy(x1).to(dev), l1, None, int(g1["seed"]), f1, j1)
    torch.cuda.synchronize()
    r["max_abs_feats_vs_reference_1000step_b2"] = float(max(np.abs(f1.cpu().numpy()[i, :n] - g1["feats"][i, :n]).max() for i, n in enumerate(l1)))
    r["max_abs_joints_vs_reference_1000step_b2"] = float(max(np.abs(j1.cpu().numpy()[i, :n] - g1["joints"][i, :n]).max() for i, n in enumerate(l1)))
    r["f64_floor_feats_joints"] = [float(g1["f64_diff_feats"]), float(g1["f64_diff_joints"])]
    e.close()
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=20, steps_offset=0, precision=prec)
    e.load_state_dict(wn, "denoiser."); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
    b64 = syn.make_batch(64, None, seed=1234, max_len=196)
    t64, x64, j64 = torch.from_numpy(b64.text_emb).to(dev), torch.randn(64, 196, 263, device=dev), torch.empty(64, 196, 22, 3, device=dev)
    f = lambda i, st: e.sample_novae(t64, x64, b64.lengths, None, 7 + i, None, j64, st.cuda_stream)
    dt = timed(f, 3, 1)
    r["ms_per_ddpm_step_bs64_T196"] = round(dt * 1e3 / 20, 3)
    r["tflops"] = round(1291.0 / (dt * 1e3 / 20) , 1)
    res[NAMES[prec]] = r
    e.close()
out["config4_novae"] = res
print(json.dumps(out))
