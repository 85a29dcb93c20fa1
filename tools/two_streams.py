"""Cluster calls in flight on two streams of one handle (max_in_flight 2), with and without the engine's cluster lane (hooks build, option "cluster_lane").
  python tools/two_streams.py [--calls 6]
Two cluster launches dispatched side by side can hold CUs while they spin on members that are not resident (2 x 192 workgroups, 256 CUs): if BOTH end up partly
resident they sit in the kernel's 200 ms wait bound, return NaN latents and the handle leaves the cluster loop at the next mldhip_numeric_status.  It takes two
dispatches within the same few microseconds: the run recorded in profiles/r05_loop_experiments.json did not hit it with the lane off (the second launch simply queued
behind the first: 6.7 ms per call) -- the lane removes the possibility and costs ~1.1 ms per call when consecutive calls alternate between streams (whole calls are
ordered: the text rows and copies of the next call no longer overlap the previous loop)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np
import torch
from mld_hip import _lib, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    mean, std = syn.make_mean_std()
    b = syn.make_batch(64, [60] * 64, seed=5)
    te, x0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
    for lane in (1, 0):
        e = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=64, max_frames=60, precision=1, max_in_flight=2)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std); e.finalize()
        e.set_option("cluster_lane", lane)
        ref = torch.empty(64, 1, 256, device=dev)
        e.sample(te, x0, b.lengths, ref)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        outs = [torch.zeros(64, 1, 256, device=dev) for _ in range(a.calls)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, l in enumerate(outs):
            e.sample(te, x0, b.lengths, l, None, None, streams[i & 1].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bad = sum(int(not torch.equal(l, ref)) for l in outs)
        nan = sum(int(torch.isnan(l).any()) for l in outs)
        print({"cluster_lane": lane, "calls": a.calls, "ms_total": round(dt * 1e3, 2), "calls_differing_from_serial": bad, "calls_with_nan": nan,
               "numeric_status": e.numeric_status(), "loop_launches_of_next_call": None}, flush=True)
        e.sample(te, x0, b.lengths, outs[0])
        torch.cuda.synchronize()
        print("   next call after the status query: loop launches", e.launch_counts()[0], "(2 = cluster loop, > 2000 = the handle left it)", flush=True)
        e.close()


if __name__ == "__main__":
    main()
