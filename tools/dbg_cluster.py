"""Debug: the cluster loop's exchange region (cluster 0) after a short call, on the simulator (--sim 1, run on the CPU) or on the GPU; --ref FILE compares
with a dump made by the other side and prints, per exchange tensor and buffer parity, the number of differing elements and where the first ones are."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from mld_hip import _lib, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--sim", type=int, default=0)
ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--wt", type=int, default=1)
ap.add_argument("--graph", type=int, default=0, choices=(0, 1))      # (2 = the memset-node clear of round 5: removed in round 6 with the engine option)
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--full", type=int, default=0)
ap.add_argument("--sync", type=int, default=1)
ap.add_argument("--seed", type=int, default=9)
ap.add_argument("--dump", type=int, default=0)
ap.add_argument("--prelat", type=int, default=0, help="latent-only cluster calls in front of the repeated ones")
ap.add_argument("--other", type=int, default=0, help="1: a second (exact-fp32) engine runs a full call between the cluster calls")
ap.add_argument("--skipfirst", type=int, default=0, help="1: no launch-family call on the cluster engine (reference from the other engine)")
ap.add_argument("--T", type=int, default=8)
ap.add_argument("--out", default="")
ap.add_argument("--ref", default="")
a = ap.parse_args()
dims = syn.ModelDims(num_layers=a.layers)
sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
if a.sim:
    import simlib
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=a.B, max_frames=8, num_inference_steps=a.steps, num_layers=a.layers, precision=1)
else:
    import torch
    e = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=a.B, max_frames=a.T, num_inference_steps=a.steps, num_layers=a.layers, precision=1, use_graph=a.graph)
e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae.")
mean, std = syn.make_mean_std()
e.load_tensor("mean", mean); e.load_tensor("std", std)
e.finalize()
b = syn.make_batch(a.B, [a.T] * a.B, seed=a.seed) if a.seed >= 0 else syn.make_batch(a.B, [a.T] * a.B)
if a.sim:
    te, x0 = b.text_emb, b.init_latents
    lat1, lat = np.zeros((a.B, 1, 256), np.float32), np.zeros((a.B, 1, 256), np.float32)
else:
    dev = torch.device("cuda:0")
    te, x0 = torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev)
    lat1, lat = torch.zeros(a.B, 1, 256, device=dev), torch.zeros(a.B, 1, 256, device=dev)
e2 = None
if a.other and not a.sim:
    e2 = _lib.Engine(device=0, max_batch=a.B, max_frames=a.T, num_inference_steps=a.steps, num_layers=a.layers, precision=0, use_graph=a.graph)
    e2.load_state_dict(sdd, "denoiser."); e2.load_state_dict(sdv, "vae."); e2.load_tensor("mean", mean); e2.load_tensor("std", std); e2.finalize()
    e2.set_option("loop_kernel", 1)
    lat2, j2 = torch.zeros(a.B, 1, 256, device=dev), torch.zeros(a.B, a.T, 22, 3, device=dev)
if a.skipfirst and e2 is not None:
    e2.sample(te, x0, b.lengths, lat1)
else:
    e.set_option("loop_kernel", 1)
    e.sample(te, x0, b.lengths, lat1)
e.set_option("loop_kernel", 4)
e.set_option("cluster_wt", a.wt)
if not a.sim and a.graph:
    e.set_option("cluster_graph", a.graph)
jj = None
if a.full and not a.sim:
    jj = torch.zeros(a.B, a.T, 22, 3, device=dev)
for _ in range(a.prelat):
    e.sample(te, x0, b.lengths, lat)
    if not a.sim:
        torch.cuda.synchronize()
        print("latent-only call: max abs", float(np.abs(lat.cpu().numpy() - lat1.cpu().numpy()).max()), flush=True)
lats_seen = []
for rep in range(a.repeat):
    if e2 is not None:
        e2.sample(te, x0, b.lengths, lat2, None, j2)
    if jj is not None:
        e.sample(te, x0, b.lengths, lat, None, jj)
    else:
        e.sample(te, x0, b.lengths, lat)
    if a.repeat > 1 and (a.sync or rep == a.repeat - 1):
        if not a.sim:
            torch.cuda.synchronize()
        d = (lat.cpu().numpy() if not a.sim else lat) - (lat1.cpu().numpy() if not a.sim else lat1)
        pc = np.abs(d).reshape(a.B, -1).max(1).reshape(-1, 8).max(1)
        pm = np.abs(d).reshape(a.B, -1).max(1)
        print("   bad motions", np.nonzero(pm > 1e-3)[0].tolist(), "bad columns of the worst motion", np.nonzero(np.abs(d).reshape(a.B, -1)[int(pm.argmax())] > 1e-3)[0].tolist()[:40])
        stb = (C.c_uint64 * 8)()
        e.lib.mldhip_profile_trace(e._h, b"den_cluster_status", 0, 8, stb, 8, 0)
        st32 = np.frombuffer(stb, dtype=np.uint32)
        print("call", rep, "max abs", float(np.abs(d).max()), "per cluster", np.array2string(pc, precision=2), "status", st32[:4].tolist(), flush=True)
        dumps = []
        for cl in range(a.B // 8 if a.dump else 0):
            bufc = (C.c_uint64 * (196608 // 2))()
            e.lib.mldhip_profile_trace(e._h, b"den_cluster_xbuf", cl, 8, bufc, 196608 // 2, 0)
            dumps.append(np.frombuffer(bufc, dtype=np.float32).copy())
        if rep == 0:
            good = dumps
        elif a.dump:
            P = 2 * 48 * 256
            for cl in range(a.B // 8):
                if pc[cl] < 1e-3:
                    continue
                for name, off, rows, cols in (("AO", 0, 48, 256), ("H1", P, 48, 256), ("Y", 2 * P, 48, 256), ("Z", 3 * P, 48, 256), ("H", 4 * P, 48, 1024)):
                    for par in range(2):
                        xa = dumps[cl][off + par * rows * cols: off + (par + 1) * rows * cols].reshape(rows, cols)
                        ra = good[cl][off + par * rows * cols: off + (par + 1) * rows * cols].reshape(rows, cols)
                        bad = np.argwhere(xa.view(np.uint32) != ra.view(np.uint32))
                        if len(bad):
                            print("  cluster", cl, name, par, "differing", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:48], "col blocks of 16:", sorted(set((bad[:, 1] // 16).tolist()))[:64], flush=True)
if not a.sim:
    torch.cuda.synchronize()
    lat1, lat = lat1.cpu().numpy(), lat.cpu().numpy()
print("cluster vs launch family: max abs", float(np.abs(lat - lat1).max()), "per cluster", np.abs(lat - lat1).reshape(a.B, -1).max(1).reshape(-1, 8).max(1) if a.B % 8 == 0 else "")
cap = 196608 // 2
buf = (C.c_uint64 * cap)()
n = e.lib.mldhip_profile_trace(e._h, b"den_cluster_xbuf", a.B, 8, buf, cap, 0)
x = np.frombuffer(buf, dtype=np.float32).copy()
if a.out:
    np.save(a.out, x)
    np.save(a.out + ".lat.npy", lat)
if a.ref:
    r = np.load(a.ref)
    rl = np.load(a.ref + ".lat.npy")
    print("latents vs ref dump: max abs", float(np.abs(lat - rl).max()))
    P = 2 * 48 * 256
    for name, off, rows, cols in (("AO", 0, 48, 256), ("H1", P, 48, 256), ("Y", 2 * P, 48, 256), ("Z", 3 * P, 48, 256), ("H", 4 * P, 48, 1024)):
        for par in range(2):
            xa = x[off + par * rows * cols: off + (par + 1) * rows * cols].reshape(rows, cols)
            ra = r[off + par * rows * cols: off + (par + 1) * rows * cols].reshape(rows, cols)
            if name == "H":
                d = xa.view(np.uint32) != ra.view(np.uint32)
                bad = np.argwhere(d)
                print(name, par, "differing words", int(d.sum()), "first", bad[:6].tolist())
            else:
                d = np.abs(xa - ra)
                bad = np.argwhere(d > 1e-3 * max(1.0, float(np.abs(ra).max())))
                print(name, par, "max abs diff %.3e (ref max %.3e)" % (float(d.max()), float(np.abs(ra).max())), "bad", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20],
                      "cols", sorted(set((bad[:, 1] // 16).tolist()))[:20])
