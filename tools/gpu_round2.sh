#!/bin/bash
# Round-2 evidence in one gpurun call (after tools/gpu_check.sh): rocprofv3 kernel stats of the single-request shape and of the
# diffusion-only variant (f32 and split-bf16), the serving-shape sweep and the arithmetic-mode A/B.  Output: gpurun_out/r02_*.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {  # $1 = output csv name, rest = command
  local name=$1; shift
  local d=$R/gpurun_out/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- "$@" > $R/gpurun_out/prof_$name.log 2>&1)
  local f=$(find $d -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r02_kernel_stats_$name.csv
  find $d -name "*kernel_trace*.csv" -delete
}
prof single_request python $R/bench.py --profile-child --precision f16x3 --coalesce 1
cat > /tmp/novae_run.py <<'PY'
import sys, os, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn
steps, prec = int(sys.argv[1]), int(sys.argv[2])
eng = _lib.Engine(device=0, max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                  scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0, precision=prec)
eng.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
mean, std = syn.make_mean_std(); eng.load_tensor("mean", mean); eng.load_tensor("std", std); eng.finalize()
b = syn.make_batch(64)
dev = torch.device("cuda:0")
text = torch.from_numpy(b.text_emb).to(dev); x0 = torch.randn(64, 196, 263, device=dev); j = torch.empty(64, 196, 22, 3, device=dev)
eng.sample_novae(text, x0, b.lengths, None, 1, None, j); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.sample_novae(text, x0, b.lengths, None, 1, None, j); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"steps": steps, "precision": prec, "ms_per_ddpm_step": dt * 1e3 / steps, "tflops": 1291.0 * steps / 1e3 / dt}))
PY
prof novae_f32 python /tmp/novae_run.py 20 0
prof novae_bf16x3 python /tmp/novae_run.py 20 1
tail -1 gpurun_out/prof_novae_f32.log gpurun_out/prof_novae_bf16x3.log 2>/dev/null | cut -c1-200
timeout 300 python tools/ab_serving.py 2>/dev/null > gpurun_out/r02_serving_ab.json
timeout 300 python tools/ab_precision.py 2>/dev/null > gpurun_out/r02_precision_ab.json
timeout 200 python tools/ab_strip_opts.py 2>/dev/null > gpurun_out/r02_strip_options_ab.json
ls -la gpurun_out/r02_* | cut -c30-200
