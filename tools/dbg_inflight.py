import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
MODE = sys.argv[1]
def load(eng, on_dev=False):
    sd = {**{"denoiser." + k: v for k, v in syn.make_denoiser_state_dict().items()}, **{"vae." + k: v for k, v in syn.make_vae_state_dict().items()}}
    sd["mean"], sd["std"] = syn.make_mean_std()
    if on_dev:
        sd = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    eng.load_state_dict(sd); eng.finalize()
    return sd
if MODE == "prealloc":
    junk = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
if MODE in ("twoengines", "twoengines_run"):
    big = _lib.Engine(device=0, max_batch=256, max_frames=196, max_in_flight=2)
    load(big)
    if MODE == "twoengines_run":
        bb = syn.make_batch(256, None, seed=5, max_len=196)
        jj = torch.empty(256, 196, 22, 3, device=dev)
        for _ in range(3):
            big.sample(torch.from_numpy(bb.text_emb).to(dev), torch.from_numpy(bb.init_latents).to(dev), bb.lengths, None, None, jj)
        torch.cuda.synchronize()
if MODE == "streams_first":
    pre = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for st in pre:
        with torch.cuda.stream(st):
            torch.zeros(10, device=dev)
    torch.cuda.synchronize()
eng = _lib.Engine(device=0, max_batch=64, max_frames=196, max_in_flight=4)
keep = load(eng, on_dev=(MODE == "devweights"))
import ctypes
def raw_streams(n):
    hip = ctypes.CDLL("libamdhip64.so")
    out = []
    for _ in range(n):
        p = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(p), 1) == 0
        out.append(torch.cuda.ExternalStream(p.value, device=dev))
    return out
pool = None
if MODE == "raw4": pool = raw_streams(4)
if MODE == "raw8_even": pool = raw_streams(8)[::2]
if MODE == "torch_skip": pool = [torch.cuda.Stream(device=dev) for _ in range(8)][::2]
if MODE == "torch_prio": pool = [torch.cuda.Stream(device=dev, priority=(-1 if i % 2 else 0)) for i in range(4)]
slots = []
for sl in range(4):
    b = syn.make_batch(64, None, seed=1234 + 1000 * sl, max_len=196)
    slots.append(dict(b=b, text=torch.from_numpy(b.text_emb).to(dev), lat0=torch.from_numpy(b.init_latents).to(dev), lat=torch.empty(64, 1, 256, device=dev),
                      feats=torch.empty(64, 196, 263, device=dev), joints=torch.empty(64, 196, 22, 3, device=dev), st=pool[sl] if pool else torch.cuda.Stream(device=dev)))
def step(i, allout):
    s = slots[i % 4]
    eng.sample(s["text"], s["lat0"], s["b"].lengths, s["lat"] if allout else None, s["feats"] if allout else None, s["joints"], s["st"].cuda_stream)
def run(n, allout):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i, allout)
    torch.cuda.synchronize(); return round(64 * n / (time.perf_counter() - t0), 1)
res = {}
allout = MODE in ("allout", "devweights")
for i in range(4): step(i, allout)
res["first_20"] = run(20, allout)
res["second_20"] = run(20, allout)
res["third_40"] = run(40, allout)
res["other_outputs_20"] = (run(8, not allout), run(20, not allout))[1]
res["back_20"] = (run(8, allout), run(20, allout))[1]
print(MODE, json.dumps(res))
