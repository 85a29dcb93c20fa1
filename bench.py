#!/usr/bin/env python
"""motions/sec of the MLD sampling hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without torchrun: bench.py re-launches itself as N ranks (python -m torch.distributed.run, 127.0.0.1);
      under torchrun (RANK / WORLD_SIZE set) it is one of the N ranks.

One "step" = one pass of the hot path over one batch: 50-step DDIM latent sampling with classifier-free guidance ->
motion-VAE decode -> (T,22,3) joints for B=64 synthetic HumanML3D-shaped prompts (config_mld_humanml3d.yaml, T=196),
inputs resident in HBM, text embeddings precomputed (the frozen CLIP encoder is outside this path).  The K steps are K
independent bs-64 requests that are all available at t = 0; the serving front end (mldhip_sample_many) coalesces
--coalesce of them into ONE reverse-diffusion chain + ONE decode (default: ceil(K / in-flight), at most 8), and
--in-flight (default 4) such calls overlap on the chip on separate HIP streams / engine workspaces.
`value_single_batch` is the same K steps issued strictly one bs-64 batch after another (per-batch latency view); the
"one call per request, four in flight" figure of round 1 is reported as `per_request_in_flight`.  Ranks are pure data parallel: weights
are broadcast once from rank 0 (one RCCL broadcast of the packed blob), every rank samples its own prompts, no data-path
collective.  Rank 0 prints ONE JSON line.

Arithmetic: the headline runs `--precision bf16x3_decode` (reverse loop, attention, norms: exact-fp32 MFMA; decoder GEMMs:
split-bf16, 3 bf16 MFMAs with fp32 accumulate) -- the fastest mode that meets the <= 1e-3 joint tolerance against the
reference (asserted by tests/test_gpu_parity.py; its measured error is in `parity`).  All-fp32 and plain bf16 are reported
as `alt_modes`, each with its measured error.

The roofline block is the dominant kernel's algorithmic FLOPs / its duration inside the DEPENDENT chain: bench.py runs a
short rocprofv3 --kernel-trace --stats child of the same workload and reads the dispatch average (`clock: rocprofv3`);
without rocprofv3 it falls back to HIP events around the real layer chain with and without that kernel (`clock:
chain_events`).  The back-to-back launch interval (r01's number) is carried beside it, never used for `frac`.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

# The steps in flight run on separate HIP streams; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
# by reference count, and with the default the bench's four streams regularly end up sharing queues with each other or with
# the engine's capture stream (measured: 6.2 k vs 7.4 k motions/s for the same code, tools/dbg_inflight.py).  Eight queues
# give every stream of this process its own.  Must be set before the HIP runtime initialises (i.e. before `import torch`).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "motion-latent-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from mld_hip import _lib  # noqa: E402
from mld_hip import synthetic as syn  # noqa: E402

METRIC = "motions/sec (50-step DDIM + VAE decode), HumanML3D bs64, 1/2/4/8 GPU"      # BASELINE.json "metric", verbatim
FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)
BATCH, FRAMES, STEPS_DDIM = 64, 196, 50
PRECISIONS = {"f32": 0, "bf16x3_decode": 1, "bf16": 2, "fp8_denoiser": 3}
DTYPE = {"f32": "f32 (exact-fp32 MFMA everywhere)",
         "bf16x3_decode": "f32 (reverse loop, attention, norms, accumulation) + split-bf16 x3 MFMA, fp32 accumulate (decoder GEMMs)",
         "bf16": "bf16 MFMA operands in every GEMM, fp32 accumulate / attention / norms / residual stream",
         "fp8_denoiser": "fp8 e4m3 MFMA operands in the reverse-loop GEMMs, split-bf16 decoder GEMMs, fp32 elsewhere"}
# profile-hook name -> (rocprofv3 kernel-name prefix, launches per sample()) at the two shapes the bench runs the loop at:
# one bs-64 request (6B = 384 rows: latency kernels, tile32.hpp) and coalesced requests (>= 768 rows: throughput kernels, strip.hpp)
KERNEL_LATENCY = {
    "den_qkv": ("void mld::gemm_tile32_kernel<32, 4, false", 9 * STEPS_DDIM),       # (<32,2> / <32,0> serve the layers after a skip linear / layer 0)
    "den_outproj": ("void mld::gemm_tile32_kernel<16, 0, false", 9 * STEPS_DDIM),
    "den_ffn1": ("void mld::gemm_tile32_kernel<32, 1, false", 9 * STEPS_DDIM),
    "den_ffn2": ("void mld::gemm_tile32_kernel<32, 0, false", 9 * STEPS_DDIM),
    "den_final": ("mld::den_final_step_kernel", STEPS_DDIM)}
KERNEL_THROUGHPUT = {   # strip.hpp template arguments: <slabs of src0, K segments, attention, PREC, ACT, column tiles per wave, waves>
    "den_qkv": ("void mld::gemm_strip_kernel<2, 1, false, 0, 0,", 9 * STEPS_DDIM),       # (<1,1,..> after a skip linear, <0,1,..> layer 0)
    "den_outproj": ("void mld::gemm_strip_kernel<0, 1, true, 0, 0,", 9 * STEPS_DDIM),
    "den_ffn1": ("void mld::gemm_strip_kernel<1, 1, false, 0, 1,", 9 * STEPS_DDIM),
    "den_ffn2": ("void mld::gemm_kernel<2, 2, 1, 2, false, true, 0, 16, false>", 9 * STEPS_DDIM),
    "den_final": ("mld::den_final_step_kernel", STEPS_DDIM)}
KERNEL_DECODE = {
    "dec_qkv": ("void mld::gemm_kernel<2, 4, 2, 2, false, true", 9), "dec_attn": ("void mld::attn_decode", 9),      # attn_decode_kernel (f32) / attn_decode_x3_kernel (the bf16-MFMA modes)
    "dec_outproj_ln": ("void mld::gemm_kernel<2, 4, 2, 4, true, true", 9), "dec_ffn1": ("void mld::gemm_kernel<2, 4, 2, 2, false, true", 9),
    "dec_ffn2_ln": ("void mld::gemm_kernel<2, 4, 2, 4, true, true", 9)}


KERNEL_DECODE_X3 = {   # split-bf16 modes: the feed-forward block is ONE launch (kernels/ffn_fused.hpp), the K = 256 staged GEMM serves QKV only
    "dec_qkv": KERNEL_DECODE["dec_qkv"], "dec_attn": KERNEL_DECODE["dec_attn"], "dec_outproj_ln": KERNEL_DECODE["dec_outproj_ln"],
    "dec_ffn": ("mld::ffn_x3_kernel", 9)}


def kernel_table(batch, precision="bf16x3_decode"):
    dec = KERNEL_DECODE_X3 if precision in ("bf16x3_decode", "fp8_denoiser") else KERNEL_DECODE
    if dec is KERNEL_DECODE_X3 and batch * 4 >= 512:     # >= 512 (sample, head) pairs: the key-blocked attention kernel (mldhip.h "flash_attn")
        dec = {**dec, "dec_attn": ("mld::attn_flash_x3_kernel", 9)}
    return {**(KERNEL_THROUGHPUT if 6 * batch >= 768 else KERNEL_LATENCY), **dec}


def algorithmic_gflop(B, T, D=256, F=1024, L=9, NF=263, steps=STEPS_DDIM):
    """SURVEY.md App. C / BASELINE.md §5 (2*MAC GEMMs + 4*S*d attention; 1-key cross-attn shortcut)."""
    lin = lambda m, k, n: 2.0 * m * k * n
    m = 3 * 2 * B
    den = L * (lin(m, D, 3 * D) + lin(m, D, D) + 4.0 * m * 3 * D + lin(m, D, F) + lin(m, F, D)) + (L - 1) / 2 * lin(m, 2 * D, D)
    md = B * T
    dec = L * (lin(md, D, 3 * D) + lin(md, D, D) + 4.0 * md * T * D + lin(md, D, F) + lin(md, F, D) + 2 * lin(B, D, D)) \
        + (L - 1) / 2 * lin(md, 2 * D, D) + lin(md, D, NF)
    return (den * steps + dec) / 1e9, den / 1e9, dec / 1e9


def source_hash():
    """Content hash of the engine sources (the GPU box has no .git): stamps profiles/*pmc* files so a stale one is refused."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "motion-latent-diffusion_amd", "csrc", "**", "*.h*"), recursive=True)) \
        + [os.path.join(REPO, "include", "mldhip.h")]
    for f in files:
        h.update(os.path.relpath(f, REPO).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def synthetic_state():
    t = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()},
         **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
    t["mean"], t["std"] = syn.make_mean_std()
    return t


def pack_and_broadcast_weights(rank, dev):
    """Rank 0 builds the synthetic checkpoint; ONE broadcast ships it (RCCL over xGMI when world > 1)."""
    from mld_hip import dp
    template = synthetic_state()
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


def make_engine(local, weights, precision, max_batch=BATCH, nfl=1, graph=True):
    eng = _lib.Engine(device=local, max_batch=max_batch, max_frames=FRAMES, use_graph=1 if graph else 0, precision=PRECISIONS[precision],
                      max_in_flight=nfl)
    eng.load_state_dict(weights)
    eng.finalize()
    return eng


def events_ms(stream, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    r = fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1), r


def time_kernel(eng, name, B, T, iters, stream):
    """Back-to-back launch interval (ms) of one named kernel, HIP events on the stream it is launched on."""
    eng.profile_kernel(name, B, T, 3, stream.cuda_stream)
    ms, flops = events_ms(stream, lambda: eng.profile_kernel(name, B, T, iters, stream.cuda_stream))
    return ms / iters, flops


def chain_marginal_us(eng, name, B, T, iters, stream):
    """In-chain cost of one den_* GEMM: HIP events around `iters` repetitions of the real dependent layer chain
    (qkv -> outproj -> ffn1 -> ffn2) minus the same chain without `name` (different kernels follow each other, as in the graph)."""
    full = ["den_qkv", "den_outproj", "den_ffn1", "den_ffn2"]

    def run(seq):
        def go():
            for _ in range(iters):
                for k in seq:
                    eng.profile_kernel(k, B, T, 1, stream.cuda_stream)
        go()
        return events_ms(stream, go)[0] / iters * 1e3
    return max(0.0, run(full) - run([k for k in full if k != name]))


def rocprof_child_stats(precision, coalesce, timeout=240):
    """rocprofv3 --kernel-trace --stats over a short single-stream run of THIS workload in a child process -> {kernel: (avg ns, calls)}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="mld_rocprof_")
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "bench", "--", sys.executable,
           os.path.abspath(__file__), "--profile-child", "--precision", precision, "--coalesce", str(coalesce)]
    try:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            return None, "no kernel_stats.csv produced"
        rows = list(csv.DictReader(open(files[0])))
        keep = os.environ.get("MLD_BENCH_KEEP_ROCPROF")     # tools/gpu_check.sh: keep the very summary the JSON line was computed from
        if keep:
            shutil.copy(files[0], keep)
        return {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in rows}, \
            "child run: bench.py --profile-child --coalesce %d (3 calls, one at a time)" % coalesce
    except Exception as ex:  # never let the profiler take the bench down
        return None, repr(ex)[:200]
    finally:
        shutil.rmtree(out, ignore_errors=True)


def cpu_baseline(seed, threads, lengths_file=None, timeout=300):
    """The oracle ("port" of the reference path, torch-CPU backend) on the host cores: one full batch (64 motions, T=196,
    50 steps) in a child process with a bounded runtime.  Returns (info, joints)."""
    out_npy = "/tmp/mld_cpu_baseline_joints_%d.npy" % os.getpid()
    cmd = [sys.executable, os.path.join(REPO, "oracle", "cpu_baseline.py"), "--batch", str(BATCH), "--frames", str(FRAMES),
           "--seed", str(seed), "--threads", str(threads), "--out", out_npy]
    if lengths_file:
        cmd += ["--lengths", lengths_file]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        return info, np.load(out_npy)
    except Exception as ex:  # timeout / parse failure: report it, never hang the bench
        return {"error": repr(ex)[:200]}, None


def run_steps(call, n, streams, single=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        call(i, single if single is not None else streams[i % len(streams)])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def bench_a2m(local, dev, warmup, steps, streams, B=256, T=60):
    """BASELINE config 5 shape (config_mld_humanact12.yaml: action condition, 15-layer denoiser, ActorVae decoder, bs=256,
    T=60) in every arithmetic mode incl. the fp8 denoiser GEMMs BASELINE.json names, each with its measured error against
    the reference-generated fixture (tests/golden/action_b256.npz: final latents).  Secondary line, never `value`."""
    dims = syn.ModelDims(num_layers=15, nfeats=150)
    sdd, sdv = syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), syn.make_actor_vae_state_dict()
    gold = np.load(os.path.join(REPO, "tests", "golden", "action_b256.npz"))
    acts, lat0, lens = syn.make_action_batch(B, nframes=T, seed=1234)      # the fixture's inputs (oracle/make_golden.py main_action)
    x0 = torch.from_numpy(lat0).to(dev)
    _, den15, _ = algorithmic_gflop(B, T, L=15, NF=150)
    _, _, dec6 = algorithmic_gflop(B, T, L=6, NF=150)
    gflop = den15 * STEPS_DDIM + dec6 - (6 - 1) / 2 * 2.0 * B * T * 512 * 256 / 1e9   # ActorVae has no skip linears
    modes = {}
    nfl = len(streams)          # the caller's streams: their hardware-queue placement is already known to be good (DESIGN.md §3 point 15)
    for prec in ("f32", "bf16x3_decode", "bf16", "fp8_denoiser"):
        eng = _lib.Engine(device=local, max_batch=B, max_frames=T, condition=_lib.COND_ACTION, nclasses=12, vae_arch=_lib.VAE_ACTOR,
                          vae_num_layers=6, num_layers=15, nfeats=150, max_in_flight=nfl, precision=PRECISIONS[prec])
        eng.load_state_dict(sdd, "denoiser.")
        eng.load_state_dict(sdv, "vae.")
        eng.finalize()
        lat = torch.empty(B, 1, 256, device=dev)
        feats = [torch.empty(B, T, 150, device=dev) for _ in range(nfl)]
        eng.sample_action(acts, x0, lens, lat, feats[0])
        torch.cuda.synchronize()
        err = float(np.abs(lat.cpu().numpy() - gold["latents"]).max())
        call = lambda i, st: eng.sample_action(acts, x0, lens, None, feats[i % nfl], st.cuda_stream)
        run_steps(call, max(warmup, nfl), streams)
        dt = run_steps(call, steps, streams)
        n1 = max(2, steps // 2)
        dt1 = run_steps(call, n1, streams, single=streams[0]) / n1
        modes[prec] = {"value": round(B * steps / dt, 1), "value_single_batch": round(B / dt1, 1), "ms_per_step_single": round(dt1 * 1e3, 3),
                       "achieved_tflops": round(gflop / 1e3 / (dt / steps), 1), "max_abs_latents_vs_reference": err}
        eng.close()
    return {"workload": "config_mld_humanact12.yaml (action-to-motion), bs=256, T=60, 50-step DDIM, CFG 7.5, ActorVae decode -> feats; "
                        "%d steps in flight; reverse loop at 6B = 1536 rows on the throughput kernels (kernels/strip.hpp)" % nfl,
            "unit": "motions/s", "algorithmic_gflop_per_batch": round(gflop, 1),
            "latents_absmax": float(np.abs(gold["latents"]).max()), "reference_vs_oracle_floor_latents": float(gold["oracle_diff_latents"]),
            "modes": modes}


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


def bench_novae(local, dev, full, streams, B=64, T=196):
    """BASELINE config 4 shape (config_novae_humanml3d.yaml: raw-motion diffusion, trans_dec denoiser d=512, bs=64, T=196,
    DDPM).  Default: 100 DDPM steps per batch (the same per-step work as the 1000-step sampler; `value` is then the
    EXTRAPOLATED 1000-step rate and says so); --full runs the real 1000 steps.  Per arithmetic mode, `nfl` batches in flight
    (own handle, stream and host thread each: a long call blocks its host thread on the hardware queue depth)."""
    import threading
    nfl = len(streams)
    steps = 1000 if full else 100
    b = syn.make_batch(B, None, seed=1234, max_len=T)
    text = torch.from_numpy(b.text_emb).to(dev)
    mean, std = syn.make_mean_std()
    weights = syn.make_novae_denoiser_state_dict()
    lin = lambda m, k, n: 2.0 * m * k * n
    m = 2 * B * T
    gf_step = (9 * (lin(m, 512, 1536) + 3 * lin(m, 512, 512) + 2 * lin(m, 512, 1024) + 4.0 * m * T * 512 + 4.0 * m * 2 * 512)
               + lin(m, 263, 512) + lin(m, 512, 263)) / 1e9
    modes = {}
    for prec in ("f32", "bf16x3_decode", "bf16"):
        engs, x0, joints = [], [], []
        for i in range(nfl):
            eng = _lib.Engine(device=local, max_batch=B, max_frames=T, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                              scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0, precision=PRECISIONS[prec])
            eng.load_state_dict(weights, "denoiser.")
            eng.load_tensor("mean", mean)
            eng.load_tensor("std", std)
            eng.finalize()
            engs.append(eng)
            x0.append(torch.randn(B, T, 263, device=dev))
            joints.append(torch.empty(B, T, 22, 3, device=dev))
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

        run(99)                              # untimed: captures each handle's step-chunk graphs
        dt = run(1234)
        ms_step = dt * 1e3 / (steps * nfl)
        modes[prec] = {"ms_per_ddpm_step": round(ms_step, 3), "achieved_tflops": round(gf_step / ms_step, 1),
                       "frac_of_fp32_mfma_peak": round(gf_step / ms_step / FP32_MFMA_PEAK_TF, 4),
                       "value": round(B / ms_step, 3), "finite": bool(all(torch.isfinite(j).all().item() for j in joints))}
        for eng in engs:
            eng.close()
    return {"workload": "config_novae_humanml3d.yaml (raw-motion diffusion, trans_dec d=512), bs=64, T=196, DDPM, CFG 7.5 -> joints; "
                        "%d batches in flight; %d DDPM steps run per batch" % (nfl, steps),
            "unit": "motions/s of the 1000-step sampler (= 64 / (1000 x ms_per_ddpm_step))", "extrapolated_from_steps": None if full else steps,
            "algorithmic_gflop_per_ddpm_step": round(gf_step, 1), "kernel_launches_per_ddpm_step": 114, "modes": modes,
            "error_vs_reference": "f32: tests/test_gpu_parity.py::test_novae_full_length_1000_steps_vs_reference_golden; every mode: "
                                  "tools/ab_precision.py -> profiles/r02_precision_ab.json"}


def profile_child(a):
    """--profile-child: the headline call shape (a.coalesce bs-64 requests per mldhip_sample_many call; 1 = a plain mldhip_sample),
    one call at a time, a few calls; run under rocprofv3 by rocprof_child_stats() and tools/gpu_pmc.sh."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = max(1, a.coalesce)
    eng = make_engine(0, synthetic_state(), a.precision, max_batch=BATCH * c)
    reqs = []
    for i in range(c):
        bt = syn.make_batch(BATCH, None, seed=1234 + 1000 * i, max_len=FRAMES)
        reqs.append(dict(text_emb=torch.from_numpy(bt.text_emb).to(dev), init_latents=torch.from_numpy(bt.init_latents).to(dev), lengths=bt.lengths,
                         joints_out=torch.empty(BATCH, FRAMES, 22, 3, device=dev)))
    for _ in range(max(1, a.steps if a.steps < 20 else 4)):
        if c == 1:
            eng.sample(reqs[0]["text_emb"], reqs[0]["init_latents"], reqs[0]["lengths"], None, None, reqs[0]["joints_out"])
        else:
            eng.sample_many(reqs)
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", choices=list(PRECISIONS)[:3], default=os.environ.get("MLD_BENCH_PRECISION", "bf16x3_decode"),
                    help="arithmetic mode of the headline (see the module docstring); the others are reported as alt_modes")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MLD_BENCH_IN_FLIGHT", "4")),
                    help="engine calls in flight per GPU: consecutive calls rotate over this many HIP streams / engine workspaces")
    ap.add_argument("--coalesce", type=int, default=int(os.environ.get("MLD_BENCH_COALESCE", "0")),
                    help="bs-64 requests coalesced into one engine call (mldhip_sample_many); 0 = ceil(steps / in-flight), at most 8; "
                         "1 = one call per request (round 1's setting)")
    ap.add_argument("--eager", action="store_true", help="disable hipGraph replay (debug)")
    ap.add_argument("--full", action="store_true", help="run the config-4 leg at its real length (1000 DDPM steps; ~1 min)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child (roofline falls back to chain events)")
    ap.add_argument("--no-alt", action="store_true", help="skip the alternate arithmetic modes and the coalesced-request measurement")
    ap.add_argument("--no-a2m", action="store_true", help="skip the secondary action-to-motion (config 5) measurement")
    ap.add_argument("--no-clip", action="store_true", help="skip timing a random-init CLIP text tower on PyTorch-ROCm")
    ap.add_argument("--no-novae", action="store_true", help="skip the secondary diffusion-only (config 4) measurement")
    ap.add_argument("--profile-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.profile_child:
        return profile_child(a)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, same flags
        import socket
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
    if world > 1 or "WORLD_SIZE" in os.environ:      # under torchrun the collective path runs at every world size, 1 included
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nfl = max(1, min(8, a.in_flight))
    coalesce = a.coalesce if a.coalesce > 0 else max(1, min(8, -(-a.steps // nfl)))
    weights, weight_bytes, bcast_s = pack_and_broadcast_weights(rank, dev)
    eng = make_engine(local, weights, a.precision, max_batch=BATCH * coalesce, nfl=nfl, graph=not a.eager)
    ident = (rank, local) + device_identity(local)
    ranks_seen = [ident]
    if dist:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, ident)

    # nfl x coalesce request slots, each with its own prompts (seeded per rank and slot) and buffers; call g of the timed
    # region serves the `coalesce` requests of group g % nfl on stream g % nfl.  Every request is one full pass of the hot
    # path over one bs-64 batch; the last call may hold fewer requests so that exactly K steps are timed.
    stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    slots = []
    for sl in range(nfl * coalesce):
        bt = syn.make_batch(BATCH, None, seed=1234 + rank + 1000 * sl, max_len=FRAMES)
        slots.append({"batch": bt, "text": torch.from_numpy(bt.text_emb).to(dev), "lat0": torch.from_numpy(bt.init_latents).to(dev),
                      "lat": torch.empty(BATCH, 1, 256, device=dev), "feats": torch.empty(BATCH, FRAMES, 263, device=dev),
                      "joints": torch.empty(BATCH, FRAMES, 22, 3, device=dev)})
    joints = slots[0]["joints"]

    def request(sl, lengths=None):
        s = slots[sl]
        return dict(text_emb=s["text"], init_latents=s["lat0"], lengths=lengths if lengths is not None else s["batch"].lengths,
                    latents_out=s["lat"], feats_out=s["feats"], joints_out=s["joints"])

    def call(e_, g, nreq, st, lengths=None):
        """one engine call: the first `nreq` requests of group g % nfl on stream st"""
        base = (g % nfl) * coalesce
        if nreq == 1 and coalesce == 1:
            s = slots[base]
            e_.sample(s["text"], s["lat0"], lengths[base] if lengths else s["batch"].lengths, s["lat"], s["feats"], s["joints"], st.cuda_stream)
        else:
            e_.sample_many([request(base + k, lengths[base + k] if lengths else None) for k in range(nreq)], st.cuda_stream)

    def issue(e_, nsteps, lengths=None):
        g, left = 0, nsteps
        while left > 0:
            n = min(coalesce, left)
            call(e_, g, n, streams[g % nfl], lengths)
            g, left = g + 1, left - n

    def step_single(e_, i, lengths=None):
        s = slots[i % len(slots)]
        e_.sample(s["text"], s["lat0"], lengths[i % len(slots)] if lengths else s["batch"].lengths, s["lat"], s["feats"], s["joints"], stream.cuda_stream)

    def timed(fn):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
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
    wsteps = max(a.warmup, 1) * coalesce * nfl   # W untimed rounds: every workspace captures every call shape of the timed region
    issue(eng, wsteps)
    if a.steps % coalesce:
        for g in range(nfl):
            call(eng, g, a.steps % coalesce, streams[g])
    for i in range(2):
        step_single(eng, i)
    dt, per_rank_s = timed(lambda: issue(eng, a.steps))
    ms_per_step = dt / a.steps * 1e3
    value = world * BATCH * a.steps / dt
    dt1 = timed(lambda: [step_single(eng, i) for i in range(a.steps)])[0]      # the same steps strictly one bs-64 batch after another
    gf_total, gf_den, gf_dec = algorithmic_gflop(BATCH, FRAMES)
    tf_job = gf_total / 1e3 / (ms_per_step * 1e-3)

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "motions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[a.precision], "data": "synthetic",
        "value_single_batch": round(world * BATCH * a.steps / dt1, 2), "ms_per_step_single_batch": round(dt1 / a.steps * 1e3, 4),
        "config": {"workload": "config_mld_humanml3d.yaml, bs=64 per step (request), T=196, 50-step DDIM, CFG 7.5, VAE decode + feats2joints; "
                               "%d requests coalesced per engine call (mldhip_sample_many), %d calls in flight per GPU on %d HIP streams "
                               "(value_single_batch: one bs-64 batch at a time)" % (coalesce, nfl, nfl),
                   "requests_per_call": coalesce, "in_flight": nfl, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "global_batch": BATCH * world, "parallelism": f"dp{world}", "graph": not a.eager, "precision": a.precision,
                   "weights": "synthetic (seeded numpy), one broadcast of %.1f MB" % (weight_bytes / 1e6),
                   "launches_per_step": eng.launch_counts()},
        "whole_job": {"algorithmic_gflop_per_batch": round(gf_total, 1), "denoise_gflop_per_step": round(gf_den, 3), "decode_gflop": round(gf_dec, 1),
                      "achieved_tflops": round(tf_job, 2), "frac_of_fp32_mfma_peak": round(tf_job / FP32_MFMA_PEAK_TF, 4),
                      "frac_of_bf16_mfma_peak": round(tf_job / BF16_MFMA_PEAK_TF, 5),
                      "note": "per GPU, amortised over the steps in flight; the reverse loop (58 % of the FLOPs) is exact fp32 in the parity modes, "
                              "so the fp32 MFMA peak is the bound that applies to it"},
    }
    per_rank_v = [BATCH * a.steps / t for t in per_rank_s]
    out["distributed"] = {"backend": "nccl (RCCL)" if dist else "none (single process)", "world_size": world,
                          "ranks_seen": [list(r) for r in ranks_seen], "distinct_devices": len({r[2] for r in ranks_seen}),
                          "weight_broadcast": {"bytes": weight_bytes, "seconds": round(bcast_s, 4), "collectives": 1 if dist else 0,
                                               "note": "one packed broadcast from rank 0 (includes host->device staging on rank 0)"},
                          "per_rank_motions_per_s": {"min": round(min(per_rank_v), 2), "max": round(max(per_rank_v), 2)},
                          "data_path_collectives": 0}
    if rank == 0:
        # ---- per-kernel table at the shape of one headline call (coalesce x 64 motions): back-to-back launch interval
        #      (HIP events on the launch stream) ...
        PB = BATCH * coalesce
        table = kernel_table(PB, a.precision)
        kern = {}
        for name, (_, cnt) in table.items():
            ms, fl = time_kernel(eng, name, PB, FRAMES, 100 if name.startswith("den") else max(6, 30 // coalesce), stream)
            kern[name] = {"interval_us": round(ms * 1e3, 2), "gflop": round(fl / 1e9, 4), "launches_per_call": cnt}
        # ---- ... and each kernel's duration inside the dependent chain: rocprofv3 dispatch averages of a child run of this call shape
        stats, where = (None, "disabled") if (a.no_rocprof or dist is not None or a.eager) else rocprof_child_stats(a.precision, coalesce)
        for name, (prefix, cnt) in table.items():
            k = kern[name]
            if stats:
                hits = [(n, v) for n, v in stats.items() if n.startswith(prefix)]
                if name.startswith("dec_") and name not in ("dec_attn", "dec_ffn"):
                    # decoder GEMMs share two templates: K = 256 vs K = 1024 differ in the KCS argument; QKV and FFN1 are one kernel (same tile, same K)
                    kcs = ", 32, false>" if name == "dec_ffn2_ln" else ", 8, false>"
                    hits = [(n, v) for n, v in hits if kcs in n]
                if hits:
                    n, (avg, calls) = max(hits, key=lambda kv: kv[1][1])
                    k["rocprof_us"], k["rocprof_kernel"], k["rocprof_calls"] = round(avg / 1e3, 2), n[:90], calls
            if name in ("den_qkv", "den_outproj", "den_ffn1", "den_ffn2"):
                k["chain_us"] = round(chain_marginal_us(eng, name, PB, FRAMES, 60, stream), 2)
            dur = k.get("rocprof_us") or k.get("chain_us") or k["interval_us"]
            k["clock"] = "rocprofv3" if "rocprof_us" in k else "chain_events" if "chain_us" in k else "interval"
            if name in ("dec_qkv", "dec_ffn1") and "dec_ffn1" in table and "rocprof_us" in k:
                # one kernel serves both: split its average by their FLOP ratio is not measurable -- report the interval clock for these two
                dur, k["clock"] = k["interval_us"], "interval (shares its rocprofv3 row with the other K = 256 decoder GEMM)"
            k["tflops"] = round(k["gflop"] / (dur * 1e-6) / 1e3, 3) if dur else 0.0
            k["share_ms"] = round(dur * cnt / 1e3, 3)
        dom = max(kern, key=lambda n: kern[n]["share_ms"])
        d = kern[dom]
        traffic, traffic_note = None, "no PMC summary (profiles/r02_pmc_traffic.json)"
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "r02_pmc_traffic.json")))
            if pmc.get("source_hash") == source_hash() and pmc.get("requests_per_call") == coalesce:
                traffic = pmc["kernels"][dom]["traffic_bytes_per_launch"]
                traffic_note = "profiles/r02_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on THIS source hash and call shape; L2<->fabric bytes per launch"
            else:
                traffic_note = "profiles/r02_pmc_traffic.json was collected on source hash %s / %s requests per call, this run is %s / %d: refused as stale" % (
                    pmc.get("source_hash"), pmc.get("requests_per_call"), source_hash(), coalesce)
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": d["tflops"], "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                           "frac": round(d["tflops"] / FP32_MFMA_PEAK_TF, 4), "traffic": traffic, "traffic_source": traffic_note,
                           "clock": d["clock"], "avg_us": d.get("rocprof_us") or d.get("chain_us") or d["interval_us"],
                           "avg_us_rocprof_dispatch": d.get("rocprof_us"), "avg_us_in_chain_events": d.get("chain_us"),
                           "avg_us_back_to_back_interval": d["interval_us"], "gflop_per_launch": d["gflop"],
                           "shape": "one headline call: %d motions, reverse loop at %d token rows, decoder at %d frame rows" % (PB, 6 * PB, PB * FRAMES),
                           "dtype_of_kernel": "f32 MFMA" if dom.startswith("den") or a.precision == "f32" else "split-bf16 x3 MFMA",
                           "rocprof": where if stats else "unavailable: %s" % where, "source_hash": source_hash()}
        out["kernels"] = kern
        cj = None
        if world == 1 and not a.no_cpu_baseline:
            # MKL/OpenMP oversubscribes badly on a 256-thread host with these small GEMMs: use <= 32 threads
            threads = min(32, os.cpu_count() or 1)
            info, cj = cpu_baseline(1234 + rank, threads)
            if cj is not None:
                out["cpu_baseline"] = {"value": round(info["motions_per_s"], 2), "unit": "motions/s", "cores": info["threads"],
                                       "kind": "port", "host_cpus": info["cores"],
                                       "sample": "1 batch of 64 motions (T=196, 50 steps) through oracle.mld_oracle (torch-CPU backend), %.1f s" % info["seconds"],
                                       "reference_modules_survey": {
                                           "value": 15.2, "unit": "motions/s", "cores": 8, "host": "survey sandbox, Xeon 2.1 GHz, MKL",
                                           "provenance": "SURVEY.md §8(d) probe: the reference's own MldDenoiser / MldVae modules + restated DDIM at B=64 "
                                                         "(4.2 s per batch); /root/reference cannot travel to the GPU box, so it is not re-timed here"}}
                issue(eng, coalesce)              # the headline call shape again: slot 0's joints come from a coalesced call
                torch.cuda.synchronize()
                err_c = float(np.abs(joints.cpu().numpy() - cj).max())
                step_single(eng, 0)
                torch.cuda.synchronize()
                out["parity"] = {"max_abs_joints_vs_oracle": max(err_c, float(np.abs(joints.cpu().numpy() - cj).max())), "tolerance": 1e-3,
                                 "precision": a.precision, "max_abs_joints_vs_oracle_coalesced_call": err_c,
                                 "max_abs_joints_vs_oracle_single_call": float(np.abs(joints.cpu().numpy() - cj).max())}
            else:
                out["cpu_baseline"] = {"value": None, "unit": "motions/s", "cores": threads, "kind": "port", "sample": str(info)}
        if world == 1 and not a.eager and not a.no_alt:
            # ---- the other arithmetic modes on the same workload and timing rule, each with its measured error
            alts = {}
            for prec in [p for p in ("f32", "bf16x3_decode", "bf16") if p != a.precision]:
                e2 = make_engine(local, weights, prec, max_batch=BATCH * coalesce, nfl=nfl)
                issue(e2, coalesce * nfl)
                if a.steps % coalesce:
                    for g in range(nfl):
                        call(e2, g, a.steps % coalesce, streams[g])
                for i in range(2):
                    step_single(e2, i)
                torch.cuda.synchronize()
                t4 = run_steps(lambda i, st: issue(e2, a.steps), 1, [None])
                n1 = max(4, a.steps // 2)
                t1 = run_steps(lambda i, st: step_single(e2, i), n1, [None]) / n1
                alt = {"value": round(BATCH * a.steps / t4, 2), "value_single_batch": round(BATCH / t1, 2), "dtype": DTYPE[prec]}
                if cj is not None:
                    step_single(e2, 0)
                    torch.cuda.synchronize()
                    alt["max_abs_joints_vs_oracle"] = float(np.abs(joints.cpu().numpy() - cj).max())
                alts[prec] = alt
                e2.close()
            out["alt_modes"] = alts
            # ---- round 1's serving shape: one engine call per request, `nfl` calls in flight (latency kernels, 384 rows per chain)
            if coalesce > 1:
                e1 = make_engine(local, weights, a.precision, max_batch=BATCH, nfl=nfl)
                one = lambda i, st: e1.sample(slots[i % len(slots)]["text"], slots[i % len(slots)]["lat0"], slots[i % len(slots)]["batch"].lengths,
                                              None, None, slots[i % len(slots)]["joints"], streams[i % nfl].cuda_stream)
                run_steps(one, 2 * nfl, [None])
                t = run_steps(one, max(a.steps, 4 * nfl), [None])
                out["per_request_in_flight"] = {"value": round(BATCH * max(a.steps, 4 * nfl) / t, 2), "unit": "motions/s", "calls_in_flight": nfl,
                                                "note": "mldhip_sample per bs-64 request, %d in flight: the headline shape of round 1" % nfl}
                e1.close()
            step_single(eng, 0)                  # slot 0 holds the headline engine's result again
            torch.cuda.synchronize()
        if world == 1 and not a.eager:
            # also a realistic length mix -- uniform in {40, 44, ..., 196}, seed 1234 (same Tmax, ragged masks)
            rng = np.random.Generator(np.random.PCG64(1234))
            mix = [[int(v) for v in rng.choice(np.arange(40, 197, 4), BATCH)] for _ in range(len(slots))]
            for ln in mix:
                ln[0] = FRAMES                                  # keep Tmax = 196 so buffers / graphs are the same
            issue(eng, coalesce * nfl, mix)
            dtm = run_steps(lambda i, st: issue(eng, a.steps, mix), 1, [None])
            lm = {"value": round(BATCH * a.steps / dtm, 2), "unit": "motions/s", "ms_per_step": round(dtm / a.steps * 1e3, 4),
                  "lengths": "uniform in {40..196 step 4}, seed 1234, Tmax 196; mean %.1f frames" % float(np.mean(mix))}
            if not a.no_cpu_baseline:
                lf = "/tmp/mld_bench_lengths_%d.json" % os.getpid()
                json.dump(mix[0], open(lf, "w"))
                info, jm = cpu_baseline(1234 + rank, min(32, os.cpu_count() or 1), lengths_file=lf)
                if jm is not None:
                    issue(eng, coalesce, mix)
                    torch.cuda.synchronize()
                    j = joints.cpu().numpy()
                    lm["max_abs_joints_vs_oracle"] = float(max(np.abs(j[i, :n] - jm[i, :n]).max() for i, n in enumerate(mix[0])))
                    lm["tolerance"] = 1e-3
            out["length_mix"] = lm
        if world == 1 and not a.eager:
            out["other_workloads"] = []
            if not a.no_a2m:
                out["other_workloads"].append(bench_a2m(local, dev, 2, max(4, a.steps // 3), streams[:2]))
            if not a.no_novae:
                out["other_workloads"].append(bench_novae(local, dev, a.full, streams[:2]))
        if world == 1 and not a.no_clip:
            try:
                te = bench_text_encoder(dev)
                te["single_batch_motions_per_s_incl_text"] = round(BATCH / (out["ms_per_step_single_batch"] + te["ms_per_128_prompts"]) * 1e3, 1)
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
