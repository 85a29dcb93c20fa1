#!/bin/bash
# GPU check of the diffusion-only variant: parity tests, then rocprofv3 kernel stats of a 20-step run at config 4's shape.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01_novae}
{
  echo "== pytest -m gpu -k 'novae or action'"; timeout 1200 python -m pytest tests -m gpu -x -q -k "novae or action or native" 2>&1 | tail -25
} 2>&1 | tee gpurun_out/check_${TAG}.log
cat > /tmp/novae_run.py <<'PY'
import sys, os, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "motion-latent-diffusion_amd"))
import torch
from mld_hip import _lib, synthetic as syn
steps = int(sys.argv[1])
eng = _lib.Engine(device=0, max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                  scheduler_type=_lib.SCHED_DDPM, num_inference_steps=steps, steps_offset=0)
eng.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
mean, std = syn.make_mean_std(); eng.load_tensor("mean", mean); eng.load_tensor("std", std); eng.finalize()
b = syn.make_batch(64)
dev = torch.device("cuda:0")
text = torch.from_numpy(b.text_emb).to(dev); x0 = torch.randn(64, 196, 263, device=dev); j = torch.empty(64, 196, 22, 3, device=dev)
eng.sample_novae(text, x0, b.lengths, None, 1, None, j); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.sample_novae(text, x0, b.lengths, None, 1, None, j); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"steps": steps, "ms_per_ddpm_step": dt * 1e3 / steps, "tflops": 1291.0 * steps / 1e3 / dt}))
PY
echo "== 20-step timing" | tee -a gpurun_out/check_${TAG}.log
timeout 600 python /tmp/novae_run.py 20 2>&1 | tail -2 | tee -a gpurun_out/check_${TAG}.log
echo "== rocprofv3" | tee -a gpurun_out/check_${TAG}.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o novae -- python /tmp/novae_run.py 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -16 "$f" | tee -a gpurun_out/check_${TAG}.log
find gpurun_out/prof_${TAG} -name "*kernel_trace*.csv" -size +20M -delete
tail -3 gpurun_out/prof_${TAG}.log
