"""N>1 path on CPU: world_size-2 gloo job (one process per "GPU") vs a single-process run."""
import os
import subprocess
import sys

import numpy as np
import pytest

import simlib
from mld_hip import dp
from mld_hip import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_covers_everything_in_order():
    for n in (0, 1, 5, 64, 513):
        for w in (1, 2, 3, 8):
            blocks = [dp.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [h - l for l, h in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_pack_state_roundtrip():
    t = {"a": np.arange(6, dtype=np.float32).reshape(2, 3), "b": np.ones(4, np.float32)}
    blob, index = dp.pack_state(t)
    assert blob.size == 10 and index[1] == ("b", (4,), 6)
    import torch
    out = dp.broadcast_state(t, t, torch.device("cpu"))          # no process group: identity
    assert np.array_equal(out["a"].numpy(), t["a"])


@pytest.mark.parametrize("world", [2])
def test_two_rank_job_matches_single_process(tmp_path, world):
    nprompts = 5
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(HERE, "dp_worker.py"), out, str(nprompts)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    # single process, whole batch
    dims = syn.ModelDims(num_layers=3)
    eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=8, max_frames=16, num_inference_steps=2, num_layers=3)
    eng.load_state_dict(syn.make_denoiser_state_dict(dims=dims), "denoiser.")
    eng.load_state_dict(syn.make_vae_state_dict(dims=dims), "vae.")
    mean, std = syn.make_mean_std()
    eng.load_tensor("mean", mean)
    eng.load_tensor("std", std)
    eng.finalize()
    b = syn.make_batch(nprompts, [16, 13, 7, 16, 9], seed=77)
    ref = np.zeros((nprompts, 16, 22, 3), np.float32)
    eng.sample(b.text_emb, b.init_latents, b.lengths, None, None, ref)
    seen = 0
    for key in got.files:
        _, lo, hi = key.split("_")
        lo, hi = int(lo), int(hi)
        j = got[key]
        for i in range(lo, hi):
            n = b.lengths[i]
            assert np.abs(j[i - lo, :n] - ref[i, :n]).max() < 1e-5      # samples are independent: same motions
            seen += 1
    assert seen == nprompts
    eng.close()


@pytest.mark.parametrize("mode,n", [("action", 3), ("novae", 3), ("action:pack", 3)])
def test_two_rank_sampler_job_other_variants_match_single_process(tmp_path, mode, n):
    """BASELINE config 5 (action-to-motion, quoted on 2 GPUs) and config 4 through the drop-in surface: mld_hip.MLD +
    DataParallelSampler on two gloo ranks (each with its own engine, weights from the one broadcast, noise pinned per prompt)
    must give the motions of a single-process run of the same prompts -- a motion may not depend on the rank it lands on."""
    import dp_models
    job_mode, mode = mode, mode.partition(":")[0]
    out = str(tmp_path / f"dp_{mode}.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", {"action": "29519", "novae": "29521"}.get(job_mode, "29523"), os.path.join(HERE, "dp_worker.py"), out, str(n), job_mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    if job_mode.endswith(":pack"):               # rank 0 says which sharding it chose: one busy rank holds all three prompts, rank 1 gets an empty shard
        assert "DataParallelSampler: pack over 1 busy rank(s)" in r.stdout, r.stdout[-800:]
    got = np.load(out)
    model, close = dp_models.build(mode, dp_models.state_template(mode), key=f"inject:dp_{mode}_single")
    try:
        kw = dp_models.job(mode, n)
        idx, ref = dp.DataParallelSampler(model, batch_size=4)(**kw)            # one process, one chunk
    finally:
        close()
    assert idx == list(range(n)) and sorted(got.files) == [f"m_{i}" for i in range(n)]
    for i in range(n):
        want = np.asarray(ref[i])
        assert got[f"m_{i}"].shape == want.shape == (kw["lengths"][i], 150) if mode == "action" else (kw["lengths"][i], 22, 3)
        assert np.abs(got[f"m_{i}"] - want).max() < 2e-5


def test_shard_planning_rules():
    """plan_shards / pack_range (mld_hip/dp.py): BASELINE config 3 (512 prompts, 8 ranks, bs 64) is spread one batch per rank when a bs-64 call runs the cluster
    loop, packed onto one rank (one 512-motion call) on an engine without that path; forced policies; packed ranges cover every prompt exactly once."""
    assert dp.plan_shards(512, 8, 64, 64, "auto", True)["policy"] == "spread"
    pk = dp.plan_shards(512, 8, 64, 512, "auto", False)
    assert pk["policy"] == "pack" and pk["busy_ranks"] == 1 and pk["prompts_per_busy_rank"] == 512
    assert dp.plan_shards(512, 8, 64, 512, "auto", True)["policy"] == "spread"          # the cluster loop makes bs-64 shards the better form
    assert dp.plan_shards(4096, 8, 64, 512, "auto", False)["policy"] == "spread"        # shards of 512 are big calls anyway
    p4 = dp.plan_shards(300, 4, 64, 128, "pack")
    assert p4["busy_ranks"] == 3 and p4["prompts_per_busy_rank"] == 128
    spans = [dp.pack_range(300, r, 4, 128) for r in range(4)]
    assert spans == [(0, 128), (128, 256), (256, 300), (300, 300)]
    assert [dp.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    # a forced "pack" never drops prompts (advisor r5): the union of the packed ranges is range(n) for every (n, world, max_batch), also when
    # n > world x max_batch -- then every rank is busy with ceil(n / world)
    for n, world, bs, mb in ((512, 1, 64, 64), (2048, 2, 64, 512), (300, 4, 64, 128), (7, 8, 64, 64), (1000, 3, 64, 256), (513, 8, 64, 64), (64, 8, 64, 2048)):
        plan = dp.plan_shards(n, world, bs, mb, "pack")
        spans = [dp.pack_range(n, r, world, plan["prompts_per_busy_rank"]) for r in range(world)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(n)), (n, world, mb, plan, spans)
        assert sum(1 for lo, hi in spans if hi > lo) == plan["busy_ranks"], (plan, spans)
