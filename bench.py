#!/usr/bin/env python
"""motions/sec of the MLD sampling hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: 50-step DDIM latent sampling with
classifier-free guidance -> motion-VAE decode -> (T,22,3) joints for B=64 synthetic HumanML3D-shaped
prompts (config_mld_humanml3d.yaml, T=196), inputs resident in HBM, text embeddings precomputed
(the frozen CLIP encoder is outside this path).  Consecutive steps rotate over --in-flight (default 4) HIP streams /
engine workspaces, so up to four independent bs-64 steps overlap on the chip; `single_stream` in the JSON is the same
K steps issued one after another.  Ranks are pure data parallel: weights are broadcast
once from rank 0 (one RCCL broadcast of the packed blob), every rank samples its own 64 prompts, no
data-path collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "motion-latent-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from mld_hip import _lib  # noqa: E402
from mld_hip import synthetic as syn  # noqa: E402

METRIC = "motions/sec (50-step DDIM + VAE decode), HumanML3D bs64, 1/2/4/8 GPU"      # BASELINE.json "metric", verbatim
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BATCH, FRAMES, STEPS_DDIM = 64, 196, 50


def algorithmic_gflop(B, T, D=256, F=1024, L=9, NF=263, steps=STEPS_DDIM):
    """SURVEY.md App. C / BASELINE.md §5 (2*MAC GEMMs + 4*S*d attention; 1-key cross-attn shortcut)."""
    lin = lambda m, k, n: 2.0 * m * k * n
    m = 3 * 2 * B
    den = L * (lin(m, D, 3 * D) + lin(m, D, D) + 4.0 * m * 3 * D + lin(m, D, F) + lin(m, F, D)) + (L - 1) / 2 * lin(m, 2 * D, D)
    md = B * T
    dec = L * (lin(md, D, 3 * D) + lin(md, D, D) + 4.0 * md * T * D + lin(md, D, F) + lin(md, F, D) + 2 * lin(B, D, D)) \
        + (L - 1) / 2 * lin(md, 2 * D, D) + lin(md, D, NF)
    return (den * steps + dec) / 1e9, den / 1e9, dec / 1e9


def pack_and_broadcast_weights(rank, world, dev):
    """Rank 0 builds the synthetic checkpoint; ONE broadcast ships it (RCCL over xGMI when world > 1)."""
    from mld_hip import dp
    template = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()},
                **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
    template["mean"], template["std"] = syn.make_mean_std()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = dp.broadcast_state(template if rank == 0 else {}, template, dev, src=0)
    torch.cuda.synchronize()
    return state, sum(v.size for v in template.values()) * 4, time.perf_counter() - t0


def device_identity(local):
    """(uuid or PCI bus id, name) of this rank's GPU -- all-gathered so the JSON proves N distinct devices took part."""
    pr = torch.cuda.get_device_properties(local)
    ident = getattr(pr, "uuid", None)
    if ident is None:
        ident = "pci:%s:%s:%s" % (getattr(pr, "pci_domain_id", "?"), getattr(pr, "pci_bus_id", "?"), getattr(pr, "pci_device_id", "?"))
    return str(ident), pr.name


def time_kernel(eng, name, B, T, iters, stream):
    """Average duration (ms) of one named kernel, HIP events on the stream it is launched on."""
    eng.profile_kernel(name, B, T, 3, stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    flops = eng.profile_kernel(name, B, T, iters, stream.cuda_stream)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters, flops


def cpu_baseline(seed, threads, timeout=420):
    """The oracle ("port" of the reference path, torch-CPU backend) on the host cores: one full batch
    (64 motions, T=196, 50 steps) in a child process with a bounded runtime.  Returns (info, joints)."""
    import subprocess
    out_npy = "/tmp/mld_cpu_baseline_joints.npy"
    cmd = [sys.executable, os.path.join(REPO, "oracle", "cpu_baseline.py"), "--batch", str(BATCH), "--frames", str(FRAMES),
           "--seed", str(seed), "--threads", str(threads), "--out", out_npy]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        return info, np.load(out_npy)
    except Exception as ex:  # timeout / parse failure: report it, never hang the bench
        return {"error": repr(ex)[:200]}, None


def bench_a2m(local, dev, stream, warmup, steps, B=256, T=60, nfl=2):
    """BASELINE config 5 shape (config_mld_humanact12.yaml: action condition, 15-layer denoiser, ActorVae decoder,
    bs=256, T=60), same timing rule, `nfl` steps in flight; a secondary line, never the headline `value`.  fp32 like the
    headline (the fp8 denoiser GEMMs BASELINE.json muses about cannot meet the parity tolerance: DESIGN.md §3 point 9, §7)."""
    eng = _lib.Engine(device=local, max_batch=B, max_frames=T, condition=_lib.COND_ACTION, nclasses=12,
                      vae_arch=_lib.VAE_ACTOR, vae_num_layers=6, num_layers=15, nfeats=150, max_in_flight=nfl)
    dims = syn.ModelDims(num_layers=15, nfeats=150)
    eng.load_state_dict(syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), "denoiser.")
    eng.load_state_dict(syn.make_actor_vae_state_dict(), "vae.")
    eng.finalize()
    slots = []
    for sl in range(nfl):
        acts, lat0, lens = syn.make_action_batch(B, nframes=T, seed=1234 + sl)
        slots.append((acts, torch.from_numpy(lat0).to(dev), lens, torch.empty(B, T, 150, device=dev), torch.cuda.Stream(device=dev)))

    def run(n, single):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            acts, x0, lens, feats, st = slots[i % nfl]
            eng.sample_action(acts, x0, lens, None, feats, (stream if single else st).cuda_stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(max(warmup, nfl), False)
    dt = run(steps, False)
    dt1 = run(steps, True) if nfl > 1 else dt
    _, den15, _ = algorithmic_gflop(B, T, L=15, NF=150)     # 15-layer skip denoiser, per step
    _, _, dec6 = algorithmic_gflop(B, T, L=6, NF=150)       # 6 decoder layers ...
    gflop = den15 * STEPS_DDIM + dec6 - (6 - 1) / 2 * 2.0 * B * T * 512 * 256 / 1e9   # ... minus the skip linears ActorVae lacks
    out = {"workload": "config_mld_humanact12.yaml (action-to-motion), bs=256, T=60, 50-step DDIM, CFG 7.5, ActorVae decode -> feats; "
                       "%d steps in flight" % nfl,
           "value": round(B * steps / dt, 2), "unit": "motions/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "single_stream_value": round(B * steps / dt1, 2), "dtype": "f32", "algorithmic_gflop_per_batch": round(gflop, 1),
           "achieved_tflops": round(gflop / 1e3 / (dt / steps), 2), "launches_per_step": eng.launch_counts()}
    eng.close()
    return out


def bench_text_encoder(dev, n_texts=2 * BATCH, iters=10):
    """the frozen CLIP ViT-L/14 text tower is OUTSIDE the measured path (it stays on PyTorch-ROCm); this
    times a random-init tower of the same architecture (123.7 M parameters; no weights or tokenizer reachable offline) on the
    2B = 128 prompts of one bs-64 CFG batch, so the metric can also be read with text encoding included."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    cfg = CLIPTextConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, projection_dim=768,
                         vocab_size=49408, max_position_embeddings=77)
    model = CLIPTextModelWithProjection(cfg).eval().to(dev)
    ids = torch.randint(0, 49407, (n_texts, 77), device=dev)
    ids[:, -1] = 49407                                      # the EOS id the pooled output is read at
    with torch.no_grad():
        for _ in range(3):
            model(input_ids=ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model(input_ids=ids)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    del model
    torch.cuda.empty_cache()
    return {"ms_per_128_prompts": round(ms, 3), "dtype": "f32", "weights": "random init (architecture of openai/clip-vit-large-patch14 text tower)",
            "backend": "PyTorch-ROCm (not part of libmldhip)"}


def bench_novae(local, dev, stream, B=64, T=196, steps=1000, nfl=2):
    """BASELINE config 4 shape (config_novae_humanml3d.yaml: raw-motion diffusion, trans_dec denoiser d=512, DDPM x1000,
    bs=64, T=196): `nfl` full 1000-step batches in flight, timed like the headline (a secondary line, never `value`).
    MFMA-bound: 1.29 TFLOP per step, noise from the in-kernel Philox stream.  A 1000-step call is 50 graph
    launches of 2 280 kernels each and blocks its host thread on the hardware queue depth, so each batch in flight gets its
    own handle, stream and host thread (a handle is used by one thread only, as the C ABI requires)."""
    import threading
    b = syn.make_batch(B, None, seed=1234, max_len=T)
    text = torch.from_numpy(b.text_emb).to(dev)
    mean, std = syn.make_mean_std()
    weights = syn.make_novae_denoiser_state_dict()
    engs, x0, joints, streams = [], [], [], []
    for i in range(nfl):
        eng = _lib.Engine(device=local, max_batch=B, max_frames=T, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                          scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0)
        eng.load_state_dict(weights, "denoiser.")
        eng.load_tensor("mean", mean)
        eng.load_tensor("std", std)
        eng.finalize()
        engs.append(eng)
        x0.append(torch.randn(B, T, 263, device=dev))
        joints.append(torch.empty(B, T, 22, 3, device=dev))
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()

    def run(seed0):
        def one(i):
            engs[i].sample_novae(text, x0[i], b.lengths, None, seed0 + i, None, joints[i], streams[i].cuda_stream)
        th = [threading.Thread(target=one, args=(i,)) for i in range(nfl)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(99)                              # untimed: captures each handle's 50 step-chunk graphs
    dt = run(1234)
    lin = lambda m, k, n: 2.0 * m * k * n
    m = 2 * B * T
    gf_step = (9 * (lin(m, 512, 1536) + 3 * lin(m, 512, 512) + 2 * lin(m, 512, 1024) + 4.0 * m * T * 512 + 4.0 * m * 2 * 512)
               + lin(m, 263, 512) + lin(m, 512, 263)) / 1e9
    out = {"workload": "config_novae_humanml3d.yaml (raw-motion diffusion, trans_dec d=512), bs=64, T=196, 1000-step DDPM, CFG 7.5 -> joints; "
                       "%d batches in flight" % nfl,
           "value": round(nfl * B / dt, 3), "unit": "motions/s", "ms_per_step": round(dt * 1e3 / nfl, 1), "steps": nfl,
           "ms_per_ddpm_step": round(dt * 1e3 / (steps * nfl), 3), "dtype": "f32", "algorithmic_gflop_per_ddpm_step": round(gf_step, 1),
           "achieved_tflops": round(gf_step * steps * nfl / 1e3 / dt, 2),
           "frac_of_fp32_mfma_peak": round(gf_step * steps * nfl / 1e3 / dt / FP32_MFMA_PEAK_TF, 4),
           "finite": bool(all(torch.isfinite(j).all().item() for j in joints)), "kernel_launches_per_ddpm_step": 114}
    for eng in engs:
        eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="disable hipGraph replay (debug)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MLD_BENCH_IN_FLIGHT", "4")),
                    help="bs-64 batches in flight per GPU: consecutive steps rotate over this many HIP streams / engine workspaces")
    ap.add_argument("--no-a2m", action="store_true", help="skip the secondary action-to-motion (config 5) measurement")
    ap.add_argument("--no-clip", action="store_true", help="skip timing a random-init CLIP text tower on PyTorch-ROCm (reported beside the metric)")
    ap.add_argument("--no-novae", action="store_true", help="skip the secondary diffusion-only (config 4, 1000-step DDPM) measurement")
    ap.add_argument("--precision", choices=["f32", "bf16x3_decode"], default=os.environ.get("MLD_BENCH_PRECISION", "f32"),
                    help="f32: exact-fp32 MFMA everywhere; bf16x3_decode: split-bf16 MFMA in the VAE-decoder GEMMs")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, same flags
        import socket
        import subprocess
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
        have = torch.cuda.device_count()
        if have < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) are visible")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {a.gpus}, or without torchrun)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    prec = {"f32": 0, "bf16x3_decode": 1}[a.precision]
    nfl = max(1, min(8, a.in_flight))
    eng = _lib.Engine(device=local, max_batch=BATCH, max_frames=FRAMES, use_graph=0 if a.eager else 1, precision=prec, max_in_flight=nfl)
    weights, weight_bytes, bcast_s = pack_and_broadcast_weights(rank, world, dev)
    ident = (rank, local) + device_identity(local)
    ranks_seen = [ident]
    if dist:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, ident)
    eng.load_state_dict(weights)
    eng.finalize()

    # `nfl` slots, each with its own prompts (seeded per rank and slot), buffers and HIP stream; step i uses slot i % nfl.
    # Four non-default streams map onto the four hardware queues HIP creates by default: measured 1/2/3/4/5 in flight =
    # 3.9k / 6.2k / 7.0k / 7.5k / 6.5k motions/s (a fifth stream shares a hardware queue and serialises behind its twin).
    # Every step is one full pass of the hot path over one bs-64 batch; steps on different streams overlap on the GPU
    # (the engine rotates its workspaces the same way), which is how a serving loop keeps the chip busy.
    stream = torch.cuda.current_stream()
    slots = []
    for sl in range(nfl):
        bt = syn.make_batch(BATCH, None, seed=1234 + rank + 1000 * sl, max_len=FRAMES)
        slots.append({"batch": bt, "text": torch.from_numpy(bt.text_emb).to(dev), "lat0": torch.from_numpy(bt.init_latents).to(dev),
                      "lat": torch.empty(BATCH, 1, 256, device=dev), "feats": torch.empty(BATCH, FRAMES, 263, device=dev),
                      "joints": torch.empty(BATCH, FRAMES, 22, 3, device=dev),
                      "stream": torch.cuda.Stream(device=dev)})
    batch, text, lat0, joints = slots[0]["batch"], slots[0]["text"], slots[0]["lat0"], slots[0]["joints"]
    mean, std = syn.make_mean_std()

    def step(i, single_stream=False):
        s = slots[i % nfl]
        eng.sample(s["text"], s["lat0"], s["batch"].lengths, s["lat"], s["feats"], s["joints"],
                   (stream if single_stream else s["stream"]).cuda_stream)

    def timed(nsteps, single_stream=False):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i, single_stream)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0           # this rank's own K steps (before waiting for the slowest rank)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_rank = [own]
        if dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            g = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(g, torch.tensor([own], device=dev, dtype=torch.float64))
            per_rank = [float(x.item()) for x in g]
        return dt, per_rank

    torch.cuda.synchronize()                     # inputs were uploaded on the default stream
    for i in range(max(a.warmup, nfl)):          # at least one call per workspace, so every graph is captured untimed
        step(i)
    dt, per_rank_s = timed(a.steps)
    ms_per_step = dt / a.steps * 1e3
    value = world * BATCH * a.steps / dt
    dt1 = timed(a.steps, single_stream=True)[0] if nfl > 1 else dt      # the same steps strictly one after another

    out = {
        "metric": METRIC, "value": round(value, 2),
        "unit": "motions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {0: "f32", 1: "f32 (reverse loop, attention, norms) + split-bf16 x3 MFMA, fp32 accumulate (decoder GEMMs)"}[prec],
        "data": "synthetic",
        "config": {"workload": "config_mld_humanml3d.yaml, bs=64 per step, T=196, 50-step DDIM, CFG 7.5, VAE decode + feats2joints; "
                               "%d steps in flight per GPU on %d HIP streams" % (nfl, nfl),
                   "in_flight": nfl, "global_batch": BATCH * world, "parallelism": f"dp{world}", "graph": not a.eager, "precision": a.precision,
                   "weights": "synthetic (seeded numpy), one broadcast of %.1f MB" % (weight_bytes / 1e6),
                   "launches_per_step": eng.launch_counts()},
    }
    per_rank_v = [BATCH * a.steps / t for t in per_rank_s]
    out["distributed"] = {"backend": "nccl (RCCL)" if dist else "none (single process)", "world_size": world,
                          "ranks_seen": [list(r) for r in ranks_seen], "distinct_devices": len({r[2] for r in ranks_seen}),
                          "weight_broadcast": {"bytes": weight_bytes, "seconds": round(bcast_s, 4), "collectives": 1 if dist else 0,
                                               "note": "one packed broadcast from rank 0 (includes host->device staging on rank 0)"},
                          "per_rank_motions_per_s": {"min": round(min(per_rank_v), 2), "max": round(max(per_rank_v), 2)},
                          "data_path_collectives": 0}
    out["single_stream"] = {"value": round(world * BATCH * a.steps / dt1, 2), "unit": "motions/s", "ms_per_step": round(dt1 / a.steps * 1e3, 4),
                            "note": "the same K steps issued on ONE stream (one batch in flight): per-batch latency"}
    if rank == 0:
        # ---- roofline: per-kernel durations with HIP events on the launch stream, weighted by launch counts
        gf_total, gf_den, gf_dec = algorithmic_gflop(BATCH, FRAMES)
        nb = 4
        fused_ffn = os.environ.get("MLDHIP_FUSED_FFN", "0") != "0"
        ffn = {"den_ffn": 9 * STEPS_DDIM} if fused_ffn else {"den_ffn1": 9 * STEPS_DDIM, "den_ffn2": 9 * STEPS_DDIM}
        per_sample = {"den_qkv": 9 * STEPS_DDIM, "den_outproj": 9 * STEPS_DDIM, **ffn, "den_final": STEPS_DDIM,
                      "dec_qkv": 9, "dec_attn": 9, "dec_outproj_ln": 9, "dec_ffn1": 9, "dec_ffn2_ln": 9}
        kern = {}
        for name, cnt in per_sample.items():
            ms, fl = time_kernel(eng, name, BATCH, FRAMES, 200 if name.startswith("den") else 30, stream)
            kern[name] = {"avg_us": round(ms * 1e3, 2), "gflop": round(fl / 1e9, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 3),
                          "launches_per_sample": cnt, "share_ms": round(ms * cnt, 3)}
        dom = max(kern, key=lambda k: kern[k]["share_ms"])
        traffic = None
        try:   # PMC counters cannot be collected from inside this process: read the committed rocprofv3 --pmc summary
            pmc = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")))["kernels"]
            traffic = pmc[dom]["traffic_bytes_per_launch"]
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": FP32_MFMA_PEAK_TF,
                           "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / FP32_MFMA_PEAK_TF, 4), "traffic": traffic,
                           "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, L2<->fabric bytes per launch)",
                           "avg_us": kern[dom]["avg_us"], "gflop_per_launch": kern[dom]["gflop"]}
        out["kernels"] = kern
        out["whole_job"] = {"algorithmic_gflop_per_batch": round(gf_total, 1), "denoise_gflop_per_step": round(gf_den, 3),
                            "decode_gflop": round(gf_dec, 1),
                            "achieved_tflops": round(gf_total / 1e3 / (ms_per_step * 1e-3), 2),
                            "frac_of_fp32_mfma_peak": round(gf_total / 1e3 / (ms_per_step * 1e-3) / FP32_MFMA_PEAK_TF, 4),
                            "note": "per GPU, amortised over the steps in flight"}
        cj = None
        if world == 1 and not a.no_cpu_baseline:
            # MKL/OpenMP oversubscribes badly on a 256-thread host with these small GEMMs: use <= 32 threads
            threads = min(32, os.cpu_count() or 1)
            info, cj = cpu_baseline(1234 + rank, threads)
            if cj is not None:
                err = float(np.abs(joints.cpu().numpy() - cj).max())
                out["cpu_baseline"] = {"value": round(info["motions_per_s"], 2), "unit": "motions/s", "cores": info["threads"],
                                       "kind": "port", "host_cpus": info["cores"],
                                       "sample": "1 batch of 64 motions (T=196, 50 steps) through oracle.mld_oracle "
                                                 "(torch-CPU backend), %.1f s" % info["seconds"]}
                out["parity"] = {"max_abs_joints_vs_oracle": err, "tolerance": 1e-3}
            else:
                out["cpu_baseline"] = {"value": None, "unit": "motions/s", "cores": threads, "kind": "port", "sample": str(info)}
        if world == 1 and prec == 0 and not a.eager:
            # alternate arithmetic mode, same workload, same timing rule (not the headline `value`)
            eng2 = _lib.Engine(device=local, max_batch=BATCH, max_frames=FRAMES, precision=1)
            eng2.load_state_dict(weights)
            eng2.finalize()
            j2 = torch.empty_like(joints)
            for _ in range(a.warmup):
                eng2.sample(text, lat0, batch.lengths, None, None, j2, stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng2.sample(text, lat0, batch.lengths, None, None, j2, stream.cuda_stream)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            alt = {"precision": "bf16x3_decode", "compare_with": "single_stream", "value": round(BATCH * a.steps / dt2, 2),
                   "ms_per_step": round(dt2 / a.steps * 1e3, 4),
                   "max_abs_joints_vs_f32_mode": float((j2 - joints).abs().max().item())}
            if cj is not None:
                alt["max_abs_joints_vs_oracle"] = float(np.abs(j2.cpu().numpy() - cj).max())
            out["alt_mode"] = alt
            eng2.close()
        if world == 1 and not a.eager:
            # also a realistic length mix -- uniform in {40, 44, ..., 196}, seed 1234 (same Tmax, ragged masks)
            rng = np.random.Generator(np.random.PCG64(1234))
            mix = [[int(v) for v in rng.choice(np.arange(40, 197, 4), BATCH)] for _ in range(nfl)]
            for ln in mix:
                ln[0] = FRAMES                                  # keep Tmax = 196 so buffers / graphs are the same
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                s_ = slots[i % nfl]
                eng.sample(s_["text"], s_["lat0"], mix[i % nfl], s_["lat"], s_["feats"], s_["joints"], s_["stream"].cuda_stream)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            out["length_mix"] = {"value": round(BATCH * a.steps / dtm, 2), "unit": "motions/s", "ms_per_step": round(dtm / a.steps * 1e3, 4),
                                 "lengths": "uniform in {40..196 step 4}, seed 1234, Tmax 196; mean %.1f frames" % float(np.mean(mix))}
        if world == 1 and not a.eager:
            out["other_workloads"] = []
            if not a.no_a2m:
                out["other_workloads"].append(bench_a2m(local, dev, stream, max(2, a.warmup), max(4, a.steps // 2),
                                                        nfl=int(os.environ.get("MLD_BENCH_A2M_IN_FLIGHT", "4"))))
            if not a.no_novae:
                out["other_workloads"].append(bench_novae(local, dev, stream))
        if world == 1 and not a.no_clip:
            try:
                te = bench_text_encoder(dev)
                te["single_stream_motions_per_s_incl_text"] = round(BATCH / (out["single_stream"]["ms_per_step"] + te["ms_per_128_prompts"]) * 1e3, 1)
                te["note"] = "text encoding of a batch can overlap the sampling of the batches already in flight; this is the strictly serial view"
                out["text_encoder"] = te
            except Exception as ex:  # transformers missing / API drift: report, never fail the bench
                out["text_encoder"] = {"error": repr(ex)[:200]}
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
