"""(GPU, round 6) Where the time of the pipelined bs-64 mode goes: `python tools/trace_pipeline.py` runs mldhip_sample_many with "many_pipeline" = 1 (10 requests per call, 3 calls)
and, with TRACE_SERIAL=1, the same requests as serial mldhip_sample calls.  Run it under `rocprofv3 --kernel-trace --output-format csv` and feed the kernel trace to
`python tools/trace_pipeline.py --parse <kernel_trace.csv>`: durations of the cluster launches, the gaps between consecutive ones, and what ran in the gaps."""
import csv, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]


def parse(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    cl = [r for r in rows if "den_cluster_kernel" in r["Kernel_Name"]]
    cl = cl[-20:]                                            # the last two calls of 10
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in cl]
    gaps, between = [], []
    for a, b in zip(cl, cl[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        gaps.append(g)
        inside = [r for r in rows if int(a["End_Timestamp"]) <= int(r["Start_Timestamp"]) < int(b["Start_Timestamp"])]
        between.append(len(inside))
    import statistics as st
    if os.environ.get("TRACE_GAP_DETAIL") and len(cl) > 3:    # what started between two consecutive cluster launches, microseconds after the first one ended
        a, b = cl[2], cl[3]
        t0 = int(a["End_Timestamp"])
        for r in rows:
            if t0 - 20000 <= int(r["Start_Timestamp"]) < int(b["Start_Timestamp"]) + 30000 and r is not a:
                print("   %+8.1f us  %6.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]), file=sys.stderr)
    gsmall = [g for g in gaps if g < 2000]                   # (the gap between two calls holds the join + host work)
    print(json.dumps({"cluster_launches": len(cl), "duration_us": {"median": round(st.median(dur), 1), "min": round(min(dur), 1), "max": round(max(dur), 1)},
                      "gap_to_next_cluster_launch_us": {"median": round(st.median(gsmall), 1), "min": round(min(gsmall), 1), "max": round(max(gsmall), 1)},
                      "kernels_started_in_a_gap_median": st.median(between), "period_us_median": round(st.median(dur) + st.median(gsmall), 1)}))


if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    parse(sys.argv[2]); sys.exit(0)
import numpy as np, torch
from mld_hip import _lib, synthetic as syn
dev = torch.device("cuda:0")
e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1, max_in_flight=2)
e.load_state_dict(syn.make_denoiser_state_dict(), "denoiser."); e.load_state_dict(syn.make_vae_state_dict(), "vae.")
m, s = syn.make_mean_std(); e.load_tensor("mean", m); e.load_tensor("std", s); e.finalize()
reqs = []
for i in range(10):
    b = syn.make_batch(64, None, seed=500 + i)
    reqs.append(dict(text_emb=torch.from_numpy(b.text_emb).to(dev), init_latents=torch.from_numpy(b.init_latents).to(dev), lengths=b.lengths,
                     latents_out=torch.empty(64, 1, 256, device=dev), joints_out=torch.empty(64, 196, 22, 3, device=dev)))
serial = bool(os.environ.get("TRACE_SERIAL"))
if not serial:
    e.set_option("many_pipeline", 1)
for _ in range(3):
    if serial:
        for q in reqs:
            e.sample(q["text_emb"], q["init_latents"], q["lengths"], q["latents_out"], None, q["joints_out"])
    else:
        e.sample_many(reqs)
    torch.cuda.synchronize()
print("done", "serial" if serial else "pipelined")
