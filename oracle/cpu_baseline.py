"""Time the CPU oracle ("port" of the reference sampling path, torch-CPU backend) on the host cores.

Run as a separate process by bench.py's cpu_baseline leg (no GPU runtime in this process, bounded by a
timeout there).  Prints one JSON line; optionally saves the joints for the parity figure.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle._paths  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--lengths", default="", help="JSON file with one length per motion (default: all --frames)")
    ap.add_argument("--device", default="cpu", help="cpu (the baseline) or cuda (bench.py eager_same_gpu: the same code on stock ATen kernels)")
    ap.add_argument("--repeat", type=int, default=1, help="timed repetitions (the minimum is reported)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from mld_hip import synthetic as syn
    from oracle import mld_oracle as O
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    ops = O.TorchOps(device=a.device)
    sync = torch.cuda.synchronize if a.device != "cpu" else (lambda: None)
    bd = O.to_backend(ops, syn.make_denoiser_state_dict())
    bv = O.to_backend(ops, syn.make_vae_state_dict())
    b = syn.make_batch(a.batch, None, seed=a.seed, max_len=a.frames)
    if a.lengths:                       # same text embeddings / start noise, ragged lengths (bench.py length_mix)
        b.lengths[:] = [int(x) for x in json.load(open(a.lengths))]
    mean, std = syn.make_mean_std()
    args = (ops.asarray(b.text_emb), ops.asarray(b.init_latents), b.lengths, ops.asarray(mean), ops.asarray(std))
    with torch.no_grad():
        O.sample(ops, bd, bv, args[0][: 4], args[1][:2], b.lengths[:2], args[3], args[4], steps=2)   # warm the allocator
        sync()
        dt = float("inf")
        for _ in range(max(1, a.repeat)):
            t0 = time.time()
            joints = O.sample(ops, bd, bv, *args, steps=a.steps)
            sync()
            dt = min(dt, time.time() - t0)
    if a.out:
        np.save(a.out, ops.to_numpy(joints))
    print(json.dumps({"seconds": dt, "motions_per_s": a.batch / dt, "threads": torch.get_num_threads(),
                      "cores": os.cpu_count(), "batch": a.batch, "frames": a.frames, "steps": a.steps, "device": a.device}))


if __name__ == "__main__":
    main()
