"""GPU parity tests: the HIP engine (through the C ABI) vs the CPU oracle and the golden vectors
produced by the reference's own modules.  Run on an MI355X:  pytest tests -m gpu

Tolerances (fp32 arithmetic both sides, different summation orders):
  single denoiser call / decode features   1e-4   (outputs are O(3))
  final latents after 50 guided steps      5e-3   (|latents| ~ 80; re-association noise is amplified by
                                                    guidance 7.5 x 50 steps -- the reference-vs-oracle floor
                                                    stored in the fixtures is ~2e-4)
  joints                                   1e-3   (the north-star tolerance)
"""
import os
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from mld_hip import _lib  # noqa: E402
from mld_hip import synthetic as syn  # noqa: E402
from oracle import mld_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _load(eng):
    eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser.")
    eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    mean, std = syn.make_mean_std()
    eng.load_tensor("mean", mean)
    eng.load_tensor("std", std)
    eng.finalize()


@pytest.fixture(scope="module")
def eng(dev):
    e = _lib.Engine(device=0, max_batch=64, max_frames=196)
    _load(e)
    yield e
    e.close()


@pytest.fixture(scope="module")
def oracle_weights():
    ops = O.NumpyOps(np.float32)
    return ops, O.to_backend(ops, syn.make_denoiser_state_dict()), O.to_backend(ops, syn.make_vae_state_dict())


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cuda(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_native_library_is_loaded(eng):
    """The parity below must come from libmldhip.so, not from anything else."""
    maps = open("/proc/self/maps").read()
    assert "libmldhip.so" in maps
    assert "libmldhip_sim" not in maps


def test_schedule_matches_oracle(eng):
    sch = O.DDIMSchedule()
    np.testing.assert_array_equal(eng.timesteps(), sch.set_timesteps(50))
    np.testing.assert_allclose(eng.alphas_cumprod(), sch.alphas_cumprod, rtol=2e-6)


@pytest.mark.parametrize("t", [981, 1, 500])
def test_denoiser_forward_vs_golden_and_oracle(eng, dev, golden_dir, oracle_weights, t):
    ops, bd, _ = oracle_weights
    g = _gold(golden_dir, "denoiser_b3.npz")
    out = torch.empty(6, 1, 256, device=dev)
    eng.denoiser_forward(_cuda(g["sample"], dev), t, _cuda(g["text_emb"], dev), 6, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = O.denoiser_forward(ops, bd, g["sample"], t, g["text_emb"])
    assert np.abs(got - ref).max() < 1e-4
    if f"out_t{t}" in g:
        assert np.abs(got - g[f"out_t{t}"]).max() < 1e-4       # the reference's own MldDenoiser


def test_denoiser_forward_full_cfg_batch(eng, dev, oracle_weights):
    ops, bd, _ = oracle_weights
    b = syn.make_batch(64)
    x = np.concatenate([b.init_latents] * 2)
    out = torch.empty(128, 1, 256, device=dev)
    eng.denoiser_forward(_cuda(x, dev), 741, _cuda(b.text_emb, dev), 128, out)
    torch.cuda.synchronize()
    ref = O.denoiser_forward(ops, bd, x, 741, b.text_emb)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4


def test_vae_decode_vs_golden(eng, dev, golden_dir):
    g = _gold(golden_dir, "vae_decode_b3.npz")
    lengths = [int(x) for x in g["lengths"]]
    feats = torch.full((3, 100, 263), float("nan"), device=dev)
    eng.vae_decode(_cuda(g["z"], dev), lengths, feats)
    torch.cuda.synchronize()
    got = feats.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - g["feats"]).max() < 1e-4               # the reference's own MldVae.decode
    assert (got[0, 50:] == 0).all()                            # padded frames zeroed


@pytest.mark.parametrize("lengths", [[1], [16], [17, 1, 16], [196, 40, 100, 3], [64] * 5])
def test_vae_decode_edge_lengths(eng, dev, oracle_weights, lengths):
    ops, _, bv = oracle_weights
    B, T = len(lengths), max(lengths)
    z = syn._rng(11, f"z{lengths}").standard_normal((B, 1, 256)).astype(np.float32)
    feats = torch.full((B, T, 263), float("nan"), device=dev)
    eng.vae_decode(_cuda(z, dev), lengths, feats)
    torch.cuda.synchronize()
    ref = O.vae_decode(ops, bv, z, lengths)
    assert np.abs(feats.cpu().numpy() - ref).max() < 1e-4


def test_feats2joints_vs_golden(eng, dev, golden_dir):
    g = _gold(golden_dir, "vae_decode_b3.npz")
    joints = torch.empty(3, 100, 22, 3, device=dev)
    eng.feats2joints(_cuda(g["feats"], dev), 3, 100, joints)
    torch.cuda.synchronize()
    assert np.abs(joints.cpu().numpy() - g["joints"]).max() < 2e-5   # reference recover_from_ric


def test_feats2joints_long_random_walk(eng, dev, oracle_weights):
    """T=196, O(1) features: exercises the wave-level prefix sums over all 64 lanes."""
    ops = oracle_weights[0]
    f = syn._rng(5, "f2j").standard_normal((4, 196, 263)).astype(np.float32)
    mean, std = syn.make_mean_std()
    joints = torch.empty(4, 196, 22, 3, device=dev)
    eng.feats2joints(_cuda(f, dev), 4, 196, joints)
    torch.cuda.synchronize()
    ref = O.feats2joints(ops, f, mean, std)
    assert np.abs(joints.cpu().numpy() - ref).max() < 1e-4


def test_ddim_step(eng, dev):
    sch = O.DDIMSchedule()
    sch.set_timesteps(50)
    e = syn._rng(1, "e").standard_normal((64, 256)).astype(np.float32)
    x = syn._rng(2, "x").standard_normal((64, 256)).astype(np.float32)
    out = torch.empty(64, 256, device=dev)
    for t in (981, 21, 1):
        eng.ddim_step(_cuda(e, dev), t, _cuda(x, dev), out, e.size)
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy() - sch.step(e, t, x)).max() < 2e-6


def _run_sample(eng, dev, b):
    B, T = len(b.lengths), max(b.lengths)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    lat = torch.empty(B, 1, 256, device=dev)
    feats = torch.empty(B, T, 263, device=dev)
    joints = torch.empty(B, T, 22, 3, device=dev)
    eng.sample(text, lat0, b.lengths, lat, feats, joints)
    torch.cuda.synchronize()
    return lat, feats, joints, (text, lat0)


def test_pipeline_demo_batch_vs_reference_golden(eng, dev, golden_dir):
    """BASELINE config 1 shape: demo/example.txt lengths [50,100,100], 50 DDIM steps."""
    g = _gold(golden_dir, "pipeline_b3.npz")
    b = syn.SyntheticBatch(g["text_emb"], g["init_latents"], [int(x) for x in g["lengths"]])
    lat, feats, joints, _ = _run_sample(eng, dev, b)
    assert np.abs(lat.cpu().numpy() - g["latents"]).max() < 5e-3
    assert np.abs(feats.cpu().numpy() - g["feats"]).max() < 2e-4
    assert np.abs(joints.cpu().numpy() - g["joints"]).max() < 1e-3


def test_pipeline_bs64_vs_reference_golden(eng, dev, golden_dir):
    """BASELINE config 2 shape: B=64, T=196 -- the benchmarked configuration."""
    g = _gold(golden_dir, "pipeline_b64.npz")
    b = syn.make_batch(64)
    lat, feats, joints, _ = _run_sample(eng, dev, b)
    assert np.abs(lat.cpu().numpy() - g["latents"]).max() < 5e-3
    assert np.abs(feats.cpu().numpy()[:, -1] - g["feats_frame_last"]).max() < 2e-4
    assert np.abs(joints.cpu().numpy() - g["joints"]).max() < 1e-3


def test_cluster_loop_bs64_vs_reference_golden_and_launch_family(dev, golden_dir):
    """The metric's literal configuration on the kernel built for it: ONE bs-64 request, T = 196, split-f16 mode -- the reverse loop is the cluster launch
    (kernels/loop_cluster.hpp: 8 clusters of 24 workgroups -- 8 column groups per token up to 64 motions --, hand-offs inside the launch), picked automatically for calls of up to 128 motions.
    Against the reference's own outputs (pipeline_b64 fixture: reference MldDenoiser / MldVae / recover_from_ric, mld.py:290-360) at the tolerances of
    test_pipeline_bs64_vs_reference_golden, against the launch-per-GEMM family of the same engine (loop_kernel 1), and: repeated calls, calls straight
    behind each other without a host sync (a full decode between two loops), write-through and plain payload stores -- all identical to the bit."""
    g = _gold(golden_dir, "pipeline_b64.npz")
    b = syn.make_batch(64)
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1)
    _load(e)
    ns = e.numeric_status()
    assert ns["probed"] == 1 and ns["loop_split_ok"] == 1, ns          # the probe ran the cluster kernel too (mldhip.hip range_probe (c))
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    lat, feats, joints = torch.empty(64, 1, 256, device=dev), torch.empty(64, 196, 263, device=dev), torch.empty(64, 196, 22, 3, device=dev)
    e.sample(text, lat0, b.lengths, lat, feats, joints)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] == 2                                    # condition rows + ONE loop launch
    l0, j0 = lat.clone(), joints.clone()
    assert np.abs(l0.cpu().numpy() - g["latents"]).max() < 5e-3
    assert np.abs(feats.cpu().numpy()[:, -1] - g["feats_frame_last"]).max() < 2e-4
    assert np.abs(j0.cpu().numpy() - g["joints"]).max() < 1e-3
    for _ in range(4):                                                  # back to back, no host sync in between
        e.sample(text, lat0, b.lengths, lat, feats, joints)
    torch.cuda.synchronize()
    assert torch.equal(lat, l0) and torch.equal(joints, j0)
    e.set_option("cluster_wt", 1)
    e.sample(text, lat0, b.lengths, lat, None, joints)
    torch.cuda.synchronize()
    assert torch.equal(lat, l0) and torch.equal(joints, j0)
    # the 12-workgroup form (what calls of more than 64 motions run on) on the same batch: another summation order in linear2 / the skip linear, same tolerances
    e.set_option("cluster_groups", 4)
    for wt in (0, 1):
        e.set_option("cluster_wt", wt)
        for _ in range(2):
            e.sample(text, lat0, b.lengths, lat, None, joints)
        torch.cuda.synchronize()
        if wt == 0:
            l4, j4 = lat.clone(), joints.clone()
        assert torch.equal(lat, l4) and torch.equal(joints, j4)
    assert np.abs(l4.cpu().numpy() - g["latents"]).max() < 5e-3 and np.abs(j4.cpu().numpy() - g["joints"]).max() < 1e-3
    print("cluster loop, 8 vs 4 column groups at bs 64: latents %.3e joints %.3e" % (float((l4 - l0).abs().max()), float((j4 - j0).abs().max())))
    assert float((l4 - l0).abs().max()) < 1e-3
    e.set_option("loop_kernel", 1)
    e.sample(text, lat0, b.lengths, lat, None, joints)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] > 2000
    dl, dj = float((lat - l0).abs().max()), float((joints - j0).abs().max())
    print("cluster loop vs launch family at bs 64: latents %.3e joints %.3e" % (dl, dj))
    assert dl < 1e-3 and dj < 2e-4
    assert e.numeric_status()["nonfinite_values"] == 0
    e.close()


def test_cluster_calls_in_flight_on_two_streams_never_run_side_by_side(dev):
    """Two cluster launches dispatched side by side would starve each other of CUs (2 x 192 workgroups spinning on members that are not resident) until the
    200 ms wait bound fails both.  The engine keeps one lane per device (engine/params.hpp ClusterLane): six bs-64 calls alternating on two streams of a handle
    with two workspaces must return exactly what the same calls return one after another, with no non-finite value and no timeout."""
    b = syn.make_batch(64, [60] * 64, seed=5)
    e = _lib.Engine(device=0, max_batch=64, max_frames=60, precision=1, max_in_flight=2)
    _load(e)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    ref_l, ref_j = torch.empty(64, 1, 256, device=dev), torch.empty(64, 60, 22, 3, device=dev)
    e.sample(text, lat0, b.lengths, ref_l, None, ref_j)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] == 2
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [(torch.full((64, 1, 256), float("nan"), device=dev), torch.full((64, 60, 22, 3), float("nan"), device=dev)) for _ in range(6)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (l, j) in enumerate(outs):
        e.sample(text, lat0, b.lengths, l, None, j, streams[i & 1].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for l, j in outs:
        assert torch.equal(l, ref_l) and torch.equal(j, ref_j)
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == 0 and ns["loop_split_ok"] == 1, ns
    assert dt < 0.15, dt                         # six calls of ~8 ms; a starved pair would sit in its 200 ms wait bound
    e.close()


def test_sample_many_pipelined_bs64_requests_are_bit_identical_and_overlap(dev):
    """"many_pipeline" 1 (round 6): mldhip_sample_many runs bs-64 requests one after the other -- the reference's shape, batch after batch (mld.py:618-672) -- with the decode
    of request k on the engine's low-priority side stream beside the cluster launch of request k + 1 (two workspaces alternate).  Eight requests with their own prompts,
    noise and (half of them) ragged lengths: every output equals the strictly serial mldhip_sample call of that request TO THE BIT, nothing non-finite, no timeout, the
    handle still on the cluster loop; and the call really overlaps: faster than the eight serial calls."""
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1, max_in_flight=2)
    _load(e)
    reqs, solo = [], []
    for i in range(8):
        b = syn.make_batch(64, "ragged" if i % 2 else None, seed=300 + i)
        T = max(b.lengths)
        text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
        lat, feats, joints = torch.empty(64, 1, 256, device=dev), torch.empty(64, T, 263, device=dev), torch.empty(64, T, 22, 3, device=dev)
        e.sample(text, lat0, b.lengths, lat, feats, joints)
        torch.cuda.synchronize()
        assert e.launch_counts()[0] == 2
        solo.append((lat, feats, joints))
        reqs.append(dict(text_emb=text, init_latents=lat0, lengths=b.lengths, latents_out=torch.full_like(lat, float("nan")),
                         feats_out=torch.full_like(feats, float("nan")) if i != 3 else None, joints_out=torch.full_like(joints, float("nan"))))

    def serial():
        for q, (lat, feats, joints) in zip(reqs, solo):
            e.sample(q["text_emb"], q["init_latents"], q["lengths"], lat, feats, joints)
    serial(); torch.cuda.synchronize()
    t_serial = 1e9
    for _ in range(2):                                                  # (best of two / three: a timing claim must not hang on one noisy run)
        t0 = time.perf_counter(); serial(); torch.cuda.synchronize(); t_serial = min(t_serial, time.perf_counter() - t0)
    e.set_option("many_pipeline", 1)
    e.sample_many(reqs); torch.cuda.synchronize()                       # (captures the loop-only and decode-only graphs of both workspaces)
    for q in reqs:
        for k in ("latents_out", "feats_out", "joints_out"):
            if q[k] is not None:
                q[k].fill_(float("nan"))
    torch.cuda.synchronize()
    t_pipe = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); e.sample_many(reqs); torch.cuda.synchronize(); t_pipe = min(t_pipe, time.perf_counter() - t0)
    for q, (lat, feats, joints) in zip(reqs, solo):
        assert torch.equal(q["latents_out"], lat) and torch.equal(q["joints_out"], joints)
        if q["feats_out"] is not None:
            assert torch.equal(q["feats_out"], feats)
    ns = e.numeric_status()
    print("bs-64 requests, 8 per call: serial %.2f ms per request, pipelined %.2f ms per request (%.0f -> %.0f motions/s)" % (
        t_serial / 8 * 1e3, t_pipe / 8 * 1e3, 512 / t_serial, 512 / t_pipe))
    assert ns["nonfinite_values"] == 0 and ns["cluster_loop"] == 1, ns
    assert t_pipe < 0.99 * t_serial, (t_pipe, t_serial)      # (half of these requests are ragged and short: their decodes are small; bench.py times the T = 196 shape: 7.98 -> 6.98 ms)
    e.set_option("many_pipeline", 0)                                    # (host-side orchestration only: the captured graphs stay)
    e.sample(reqs[0]["text_emb"], reqs[0]["init_latents"], reqs[0]["lengths"], solo[0][0], solo[0][1], solo[0][2])
    torch.cuda.synchronize()
    assert torch.equal(solo[0][0], reqs[0]["latents_out"])
    e.close()
    # the same orchestration without graphs (use_graph = 0: eager issue of the two halves on the two streams)
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1, max_in_flight=2, use_graph=0)
    _load(e)
    outs = []
    for q in reqs[:3]:
        lat, joints = torch.empty_like(q["latents_out"]), torch.empty_like(q["joints_out"])
        e.sample(q["text_emb"], q["init_latents"], q["lengths"], lat, None, joints)
        outs.append((lat, joints))
    torch.cuda.synchronize()
    e.set_option("many_pipeline", 1)
    for q in reqs[:3]:
        q["latents_out"].fill_(float("nan")); q["joints_out"].fill_(float("nan"))
    e.sample_many([dict(q, feats_out=None) for q in reqs[:3]])
    torch.cuda.synchronize()
    for q, (lat, joints) in zip(reqs[:3], outs):
        assert torch.equal(q["latents_out"], lat) and torch.equal(q["joints_out"], joints)
    assert e.numeric_status()["nonfinite_values"] == 0
    e.close()


def test_sample_many_pipelined_repeated_mixed_calls_stay_bit_identical(dev):
    """The three-stream schedule of "many_pipeline" (prep / caller / side stream, two workspaces) under repetition and mixed use: requests of 16 .. 128 motions (one
    cluster launch each: inputs and condition rows staged on the prep stream beside the previous launch), a call that also holds a 200-motion request (two launches: the
    whole call falls back to the two-stream form), plain mldhip_sample calls in between (the workspaces change hands between entry points), all on a stream of the
    caller's own.  Every output of every round equals the request's serial mldhip_sample result to the bit; nothing non-finite, no timeout."""
    e = _lib.Engine(device=0, max_batch=256, max_frames=196, precision=1, max_in_flight=2)
    _load(e)
    sizes = [64, 16, 128, 64, 40, 64, 200]
    reqs, solo = [], []
    for i, B in enumerate(sizes):
        b = syn.make_batch(B, "ragged" if i % 2 else None, seed=900 + i)
        T = max(b.lengths)
        text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
        lat, joints = torch.empty(B, 1, 256, device=dev), torch.empty(B, T, 22, 3, device=dev)
        e.sample(text, lat0, b.lengths, lat, None, joints)
        solo.append((lat, joints))
        reqs.append(dict(text_emb=text, init_latents=lat0, lengths=b.lengths, latents_out=torch.empty_like(lat), feats_out=None, joints_out=torch.empty_like(joints)))
    torch.cuda.synchronize()
    e.set_option("many_pipeline", 1)
    st = torch.cuda.Stream()
    scratch = (torch.empty_like(solo[1][0]), torch.empty_like(solo[1][1]))
    for it in range(9):
        for q in reqs:
            q["latents_out"].fill_(float("nan")); q["joints_out"].fill_(float("nan"))
        torch.cuda.synchronize()
        if it % 3 == 0:
            e.sample_many(reqs[:6], st.cuda_stream)
            done = range(6)
        elif it % 3 == 1:
            e.sample_many(reqs[:3], st.cuda_stream)
            e.sample(reqs[1]["text_emb"], reqs[1]["init_latents"], reqs[1]["lengths"], scratch[0], None, scratch[1], st.cuda_stream)
            e.sample_many(reqs[3:6], st.cuda_stream)
            done = range(6)
        else:
            e.sample_many([reqs[0], reqs[6], reqs[2], reqs[4]], st.cuda_stream)
            done = (0, 6, 2, 4)
        st.synchronize()
        for i in done:
            assert torch.equal(reqs[i]["latents_out"], solo[i][0]), (it, i)
            assert torch.equal(reqs[i]["joints_out"], solo[i][1]), (it, i)
        if it % 3 == 1:
            assert torch.equal(scratch[0], solo[1][0]) and torch.equal(scratch[1], solo[1][1])
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == 0 and ns["cluster_loop"] == 1, ns
    e.close()


def test_cluster_calls_beside_a_foreign_kernel_stream(dev):
    """The cluster launch needs its workgroups resident together; kernels of OTHER streams beside it can only delay it -- they end.  bs-64 calls while
    a foreign stream keeps the chip busy with (a) large torch matmuls (every CU, LDS-heavy workgroups) and (b) a CLIP-sized transformer layer stack -- what the next
    batch's text encoder would be: latents and joints identical to the quiet run to the bit, nothing non-finite, no sticky timeout, the handle stays on the cluster loop."""
    b = syn.make_batch(64, [120] * 64, seed=77)
    e = _lib.Engine(device=0, max_batch=64, max_frames=120, precision=1)
    _load(e)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    ref_l, ref_j = torch.empty(64, 1, 256, device=dev), torch.empty(64, 120, 22, 3, device=dev)
    e.sample(text, lat0, b.lengths, ref_l, None, ref_j)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] == 2
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(4096, 4096, device=dev); w = torch.randn(4096, 4096, device=dev)
    layer = torch.nn.TransformerEncoderLayer(768, 12, 3072, batch_first=True).to(dev).eval()
    tok = torch.randn(64, 77, 768, device=dev)
    for name in ("matmul", "clip_like"):
        stop_after = 40
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(stop_after):                                   # ~100+ ms of foreign work queued on the side stream, running while the calls below are issued
                if name == "matmul":
                    a = torch.tanh(a @ w * 1e-2)
                else:
                    for _ in range(12):
                        tok = layer(tok)
        outs = []
        t0 = time.perf_counter()
        for _ in range(4):
            l, j = torch.full_like(ref_l, float("nan")), torch.full_like(ref_j, float("nan"))
            e.sample(text, lat0, b.lengths, l, None, j)
            outs.append((l, j))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for l, j in outs:
            assert torch.equal(l, ref_l) and torch.equal(j, ref_j), name
        ns = e.numeric_status()
        print("cluster calls beside a foreign %s stream: 4 calls + the foreign work in %.1f ms, numeric %s" % (name, dt * 1e3, {k: ns[k] for k in ("nonfinite_values", "cluster_loop")}))
        assert ns["nonfinite_values"] == 0 and ns["cluster_loop"] == 1, (name, ns)
    e.close()


def test_cluster_lane_belongs_to_one_process_tree_per_device(dev, tmp_path):
    """Advisor r5: two PROCESSES interleaving cluster launches on one GPU would each end partly resident and run every wait into its bound.  The first process that creates
    a handle on a device holds an advisory lock (/tmp/mldhip_cluster_lane_<pci bus id>.lock); here this test process is the owner (cluster_loop = 1), a child it supervises
    is part of the same tenant (1 as well: bench.py's rocprofv3 child must keep the path it profiles), and an ORPHAN -- double fork, re-parented away from this process tree,
    the stand-in for another rank or tenant on the same GPU -- gets cluster_loop = 3 and serves a bs-64 call on the launch family (2 052 launches), correctly."""
    import subprocess
    import sys
    e = _lib.Engine(device=0, max_batch=8, max_frames=16, precision=1)
    _load(e)
    assert e.numeric_status()["cluster_loop"] == 1
    prog = (
        "import sys, json, numpy as np, torch\n"
        "sys.path[:0] = %r\n"
        "from mld_hip import _lib, synthetic as syn\n"
        "e = _lib.Engine(device=0, max_batch=8, max_frames=16, precision=1)\n"
        "e.load_state_dict(syn.make_denoiser_state_dict(), 'denoiser.'); e.load_state_dict(syn.make_vae_state_dict(), 'vae.')\n"
        "m, s = syn.make_mean_std(); e.load_tensor('mean', m); e.load_tensor('std', s); e.finalize()\n"
        "b = syn.make_batch(8, [16] * 8, seed=3)\n"
        "dev = torch.device('cuda:0'); lat = torch.zeros(8, 1, 256, device=dev)\n"
        "e.sample(torch.from_numpy(b.text_emb).to(dev), torch.from_numpy(b.init_latents).to(dev), b.lengths, lat); torch.cuda.synchronize()\n"
        "json.dump({'cluster_loop': e.numeric_status()['cluster_loop'], 'launches': e.launch_counts()[0], 'lat': lat.cpu().numpy().ravel()[:64].tolist()}, open(sys.argv[1], 'w'))\n"
    ) % ([p for p in sys.path if p],)
    script = tmp_path / "tenant.py"
    script.write_text(prog)
    child_out, orphan_out = tmp_path / "child.json", tmp_path / "orphan.json"
    subprocess.run([sys.executable, str(script), str(child_out)], check=True, timeout=300)
    # the orphan: an intermediate process forks the tenant and exits at once, so the tenant is re-parented (to init / a subreaper outside this process's ancestry)
    launcher = "import os, sys\nif os.fork() == 0:\n    os.setsid()\n    os.execv(sys.executable, [sys.executable, %r, %r])\nos._exit(0)\n" % (str(script), str(orphan_out))
    subprocess.run([sys.executable, "-c", launcher], check=True, timeout=60)
    t0 = time.time()
    while not orphan_out.exists() and time.time() - t0 < 300:
        time.sleep(0.5)
    time.sleep(0.5)
    import json
    child, orphan = json.load(open(child_out)), json.load(open(orphan_out))
    print("cluster lane: owner 1, supervised child", child["cluster_loop"], child["launches"], "orphan", orphan["cluster_loop"], orphan["launches"])
    assert child["cluster_loop"] == 1 and child["launches"] == 2
    assert orphan["cluster_loop"] == 3 and orphan["launches"] > 2000
    assert np.abs(np.array(child["lat"]) - np.array(orphan["lat"])).max() < 2e-3        # the same motions on another loop family (|x| ~ 80)
    e.close()


def test_cluster_loop_bounded_waits_and_fallback_on_gpu(dev):
    """The bounded waits of the cluster launch on hardware (hooks build of the library, option "cluster_inject": one member of every cluster never raises its first
    flag, wait bound 2 ms): the call returns instead of hanging, its latents are NaN and counted, mldhip_numeric_status takes the handle off the cluster loop, the
    next call (launch family) equals the exact-fp32 engine within the split tolerance; loop_kernel 4 re-arms the cluster loop and the result is the usual one."""
    if not os.path.exists(_lib.HOOKS_LIB):
        pytest.skip("hooks build of the library not present (make -C motion-latent-diffusion_amd/csrc hooks)")
    b = syn.make_batch(64, [40] * 64, seed=31)
    e = _lib.Engine(lib=_lib.hooks_library(), device=0, max_batch=64, max_frames=40, precision=1)
    _load(e)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    lat = torch.zeros(64, 1, 256, device=dev)
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    good = lat.clone()
    assert e.launch_counts()[0] == 2 and not torch.isnan(good).any()
    e.set_option("cluster_inject", 1 + 2)                    # member (token 0, head 2) of every cluster
    t0 = time.perf_counter()
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert dt < 2.0, dt                                      # graph capture + a 2 ms bound, not a hang
    assert torch.isnan(lat).all()
    # self-healing (round 6): the kernel also set the handle's pinned host word; the NEXT sample call reads it first and runs on the launch family -- the caller has NOT
    # polled mldhip_numeric_status in between
    e.sample(text, lat0, b.lengths, lat)                     # the handle has left the cluster loop
    torch.cuda.synchronize()
    assert e.launch_counts()[0] > 2000 and float((lat - good).abs().max()) < 1e-3
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == 64 * 256 and ns["cluster_loop"] == 2, ns
    e.set_option("cluster_inject", 0)
    e.set_option("loop_kernel", 4)
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] == 2 and torch.equal(lat, good)
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == 0 and ns["cluster_loop"] == 1, ns
    # entry check (round 6): a stale epoch in a polled word fails the launch (NaN, counted) instead of being consumed as "ready"; the next call has healed
    e.set_option("cluster_stale", 1)                         # (the stale word sits in cluster 0's lines: its 8 motions are poisoned for sure, the other clusters' if they were
    e.sample(text, lat0, b.lengths, lat)                     #  still waiting when the status word went up)
    torch.cuda.synchronize()
    assert torch.isnan(lat[:8]).all()
    e.set_option("cluster_stale", 0)
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] > 2000 and float((lat - good).abs().max()) < 1e-3
    ns = e.numeric_status()
    assert ns["nonfinite_values"] >= 8 * 256 and ns["nonfinite_values"] % 2048 == 0 and ns["cluster_loop"] == 2, ns
    e.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_second_weight_family_vs_reference_golden(dev, golden_dir, prec):
    """A second family of weights (mld_hip.synthetic.trained_like: LayerNorm gains ~ N(1, 0.3), heavy-tailed weight rows, a small final gain) against the REFERENCE
    modules' outputs on those weights (tests/golden/pipeline_b8_trainedlike.npz, oracle/make_golden_trainedlike.py; B = 8 ragged, every frame kept): exact-fp32
    engine, and the split-f16 engine on the cluster loop and on the launch family.  The tolerances are the first family's; the range probe's verdict is checked."""
    g = _gold(golden_dir, "pipeline_b8_trainedlike.npz")
    lengths = [int(x) for x in g["lengths"]]
    b = syn.make_batch(8, lengths, seed=4321, max_len=64)
    e = _lib.Engine(device=0, max_batch=8, max_frames=64, precision=prec)
    e.load_state_dict(syn.trained_like(syn.make_denoiser_state_dict()), "denoiser.")
    e.load_state_dict(syn.trained_like(syn.make_vae_state_dict(), seed=12), "vae.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    if prec == 1:
        ns = e.numeric_status()
        assert ns["probed"] == 1 and ns["loop_split_ok"] == 1 and ns["decode_split_ok"] == 1, ns
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    lat, feats, joints = torch.empty(8, 1, 256, device=dev), torch.empty(8, 64, 263, device=dev), torch.empty(8, 64, 22, 3, device=dev)
    for lk in ((0, 1) if prec == 1 else (0,)):
        e.set_option("loop_kernel", lk)
        e.sample(text, lat0, b.lengths, lat, feats, joints)
        torch.cuda.synchronize()
        assert (e.launch_counts()[0] == 2) == (prec == 1 and lk == 0)
        el, ef, ej = (float(np.abs(a.cpu().numpy() - g[k]).max()) for a, k in ((lat, "latents"), (feats, "feats"), (joints, "joints")))
        print("second weight family, precision %d, loop_kernel %d: latents %.3e feats %.3e joints %.3e" % (prec, lk, el, ef, ej))
        assert el < 5e-3 and ef < 1e-3 and ej < 1e-3
    e.close()


@pytest.mark.parametrize("case", ["small", "large"])
def test_small_and_large_latent_regimes_vs_reference_golden(dev, golden_dir, case):
    """The regimes no other fixture reaches (tests/golden/pipeline_b8_latent_scales.npz = the REFERENCE modules' outputs, oracle/make_golden_latent_scales.py):
    |latent| max 7.5 -- the decoder's self-attention is no longer drowned by the per-sample cross-attention vector -- and |latent| max 312 -- the loop's split-f16 operands
    at 4x the magnitude of every other fixture.  Exact-fp32 engine and split-f16 engine (cluster loop and launch family), the probe's verdict, nothing non-finite.
    Joints and features: the contract's 1e-3; latents: 5e-3 per |x| of 80, as elsewhere."""
    g = _gold(golden_dir, "pipeline_b8_latent_scales.npz")
    sdd, sdv, b = syn.latent_scale_case(case)
    mean, std = syn.make_mean_std()
    lat_ref = g[case + "_latents"]
    tol_l = 5e-3 * max(1.0, float(np.abs(lat_ref).max()) / 80.0)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    for prec in (0, 1):
        e = _lib.Engine(device=0, max_batch=8, max_frames=64, precision=prec)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std)
        e.finalize()
        lat, feats, joints = torch.empty(8, 1, 256, device=dev), torch.empty(8, 64, 263, device=dev), torch.empty(8, 64, 22, 3, device=dev)
        for lk in ((0, 1, 3) if prec == 1 else (0,)):
            e.set_option("loop_kernel", lk)
            e.sample(text, lat0, b.lengths, lat, feats, joints)
            torch.cuda.synchronize()
            el, ef, ej = (float(np.abs(a.cpu().numpy() - g[case + "_" + k]).max()) for a, k in ((lat, "latents"), (feats, "feats"), (joints, "joints")))
            print("%s latents (|x| max %.1f), precision %d, loop_kernel %d: latents %.3e feats %.3e joints %.3e" % (case, np.abs(lat_ref).max(), prec, lk, el, ef, ej))
            assert el < tol_l and ef < 1e-3 and ej < 1e-3, (case, prec, lk, el, ef, ej)
        ns = e.numeric_status()
        assert ns["nonfinite_values"] == 0, ns
        if prec == 1:
            print("  probe:", {k: ns[k] for k in ("loop_split_ok", "decode_split_ok", "probe_err_loop", "probe_err_decode")})
        e.close()


def test_every_feature_frame_of_the_bs64_request_vs_reference_golden(dev, golden_dir):
    """MldVae.decode's features of the bs-64 pipeline -- all 196 frames of every 4th motion (tests/golden/pipeline_b64_feats.npz, the reference's own
    run, oracle/make_golden_b64_feats.py), not only the last frame -- from the split-f16 engine's full call (cluster loop) and from the exact-fp32 engine."""
    g, gf = _gold(golden_dir, "pipeline_b64.npz"), _gold(golden_dir, "pipeline_b64_feats.npz")
    motions = [int(m) for m in gf["motions"]]
    b = syn.make_batch(64)
    for prec in (0, 1):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=prec)
        _load(e)
        lat, feats, joints, _ = _run_sample(e, dev, b)
        ef = float(np.abs(feats.cpu().numpy()[motions] - gf["feats"]).max())
        ej = float(np.abs(joints.cpu().numpy() - g["joints"]).max())
        print("bs-64 request, every feature frame of 16 motions, precision %d: feats %.3e joints (all motions, all frames) %.3e" % (prec, ef, ej))
        assert ef < 1e-4 and ej < 5e-4
        e.close()


@pytest.mark.parametrize("B", [11, 128, 197])
def test_cluster_loop_ragged_and_two_clusters_per_xcd_vs_oracle(dev, B):
    """11 motions (two clusters, the second with three live motions), 128 motions (16 clusters = two per XCD, 192 workgroups) and 197 motions (two launches
    one after the other: 128 on 12-workgroup clusters + 69 ragged, sharing the exchange regions): latents of the cluster launch(es) against the CPU oracle
    (torch fp32) and against the launch family."""
    b = syn.make_batch(B, [40] * B, seed=21)
    e = _lib.Engine(device=0, max_batch=B, max_frames=40, precision=1)
    _load(e)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    lat = torch.empty(B, 1, 256, device=dev)
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    assert e.launch_counts()[0] == (2 if B <= 128 else 3)
    ops = O.TorchOps("float32")
    ref = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, syn.make_denoiser_state_dict()), ops.asarray(b.text_emb), ops.asarray(b.init_latents), 7.5, 50, 4))
    err = float(np.abs(lat.cpu().numpy() - ref).max())
    l0 = lat.clone()
    e.set_option("loop_kernel", 1)
    e.sample(text, lat0, b.lengths, lat)
    torch.cuda.synchronize()
    print("cluster loop, %d motions: latents vs oracle %.3e, vs launch family %.3e" % (B, err, float((lat - l0).abs().max())))
    assert err < 5e-3 and float((lat - l0).abs().max()) < 1e-3
    e.close()


def test_graph_replay_is_bit_identical_and_matches_eager(eng, dev):
    b = syn.make_batch(8, "ragged", seed=99)
    B, T = 8, max(b.lengths)
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    outs = []
    joints = torch.empty(B, T, 22, 3, device=dev)
    feats = torch.empty(B, T, 263, device=dev)
    for _ in range(3):                      # 1st call captures, 2nd/3rd replay the same hipGraphExec
        joints.fill_(float("nan"))
        eng.sample(text, lat0, b.lengths, None, feats, joints)
        torch.cuda.synchronize()
        outs.append(joints.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    eager = _lib.Engine(device=0, max_batch=8, max_frames=196, use_graph=0)
    _load(eager)
    j2 = torch.empty(B, T, 22, 3, device=dev)
    eager.sample(text, lat0, b.lengths, None, None, j2)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], j2)
    eager.close()


def test_samples_are_independent(eng, dev):
    """Data-parallel premise: a motion's result does not depend on its batch mates."""
    b = syn.make_batch(6, [30, 44, 52, 12, 60, 60], seed=5)
    _, _, joints, _ = _run_sample(eng, dev, b)
    sub = syn.SyntheticBatch(np.concatenate([b.text_emb[:1], b.text_emb[8:9]]), b.init_latents[2:3], [52])
    _, _, j1, _ = _run_sample(eng, dev, sub)
    assert np.abs(joints[2, :52].cpu().numpy() - j1[0].cpu().numpy()).max() < 1e-4


def test_error_behaviour(eng, dev):
    b = syn.make_batch(2, [10, 10])
    text, lat0 = _cuda(b.text_emb, dev), _cuda(b.init_latents, dev)
    with pytest.raises(_lib.MldHipError):
        eng.sample(text, lat0, [10, 500], None, None, None)        # > max_frames
    with pytest.raises(_lib.MldHipError):
        eng.sample(text, lat0, [10, 0], None, None, None)          # empty motion
    with pytest.raises(_lib.MldHipError):
        eng.denoiser_forward(lat0, 1000, text, 2, lat0)            # timestep out of range
    fresh = _lib.Engine(device=0, max_batch=2, max_frames=16)
    with pytest.raises(_lib.MldHipError):
        fresh.sample(text, lat0, [10, 10], None, None, None)       # before finalize
    fresh.load_tensor("mean", np.zeros(263, np.float32))
    with pytest.raises(_lib.MldHipError):
        fresh.finalize()                                           # std missing: strict within a group
    with pytest.raises(_lib.MldHipError):
        fresh.load_tensor("denoiser.encoder.norm.weight", np.zeros(255, np.float32))   # wrong shape
    fresh.close()


def test_mld_module_surface_on_gpu(dev):
    """The reference-shaped Python surface (YAML targets -> drop-in modules -> MLD.forward) on the real engine."""
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder

    cfg = C.load_config()
    E.configure(max_batch=8, max_frames=196)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev).eval()
    assert model.fused
    texts = ["a man kicks with something or someone with his left leg.", "A person is skipping rope.", "a person walks backward slowly."]
    lengths = [50, 100, 100]                                   # demo/example.txt
    lat0 = _cuda(syn.make_batch(3, lengths).init_latents, dev)
    joints = model({"text": texts, "length": lengths}, init_latents=lat0)
    assert [tuple(j.shape) for j in joints] == [(50, 22, 3), (100, 22, 3), (100, 22, 3)] and not joints[0].is_cuda
    ops = O.NumpyOps(np.float32)
    emb = model.text_encoder([""] * 3 + texts)
    mean, std = syn.make_mean_std()
    jr = O.sample(ops, O.to_backend(ops, syn.make_denoiser_state_dict()), O.to_backend(ops, syn.make_vae_state_dict()),
                  emb.cpu().numpy(), lat0.cpu().numpy(), lengths, mean, std)
    for i, n in enumerate(lengths):
        assert np.abs(joints[i].numpy() - jr[i, :n]).max() < 1e-3
    # per-op drop-ins in the reference's own loop (mld.py:290-360) agree with the fused graph
    z = model._diffusion_reverse(emb, lengths, init_latents=lat0)
    feats = model.vae.decode(z.contiguous(), lengths)
    j2 = model.feats2joints(feats).cpu().numpy()
    for i, n in enumerate(lengths):
        assert np.abs(j2[i, :n] - joints[i].numpy()).max() < 1e-3
    E.drop_engines()


def test_split_f16_decode_mode_meets_the_joint_tolerance(dev, golden_dir):
    """precision = F16X3: the decoder on split-f16 MFMAs, every kernel choice: the row-strip GEMMs + register-direct feed-forward
    kernel ("strip_gemm" = 1, "ffn_strip" = 6; the default 1 picks 64-row strips at this size), the 64- and 48-row strips, and the LDS-staged tiles of gemm.hpp; against
    the reference's own features / joints (pipeline_b64 fixture) and its ragged MldVae.decode fixture; the three builds agree to
    fp32-rounding class differences."""
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1)
    _load(e)
    g = _gold(golden_dir, "pipeline_b64.npz")
    gd = _gold(golden_dir, "vae_decode_b3.npz")
    b = syn.make_batch(64)
    feats_by = {}
    for sg, fs, tail in ((1, 6, 1), (1, 4, 1), (1, 3, 1), (1, 3, 0), (0, 0, 1)):      # (1, 3, 1): the one-launch decoder tail ("dec_tail")
        e.set_option("strip_gemm", sg)
        e.set_option("ffn_strip", fs)
        e.set_option("dec_tail", tail)
        lat, feats, joints, _ = _run_sample(e, dev, b)
        err_f = np.abs(feats.cpu().numpy()[:, -1] - g["feats_frame_last"]).max()
        err_j = np.abs(joints.cpu().numpy() - g["joints"]).max()
        print("f16x3 decode (strip_gemm %d, ffn_strip %d): feats err %.3e joints err %.3e" % (sg, fs, err_f, err_j))
        assert err_f < 1e-4 and err_j < 5e-4
        f3 = torch.full((3, 100, 263), float("nan"), device=dev)
        e.vae_decode(_cuda(gd["z"], dev), [int(x) for x in gd["lengths"]], f3)
        torch.cuda.synchronize()
        assert np.abs(f3.cpu().numpy() - gd["feats"]).max() < 1e-4
        feats_by[(sg, fs)] = feats.clone()
    assert 0 < (feats_by[(1, 6)] - feats_by[(0, 0)]).abs().max().item() < 1e-4      # different kernels really ran; same arithmetic class
    e.close()


def test_decoder_half_qkv_opt_in_form_probe_and_parity(dev, golden_dir):
    """"dec_half" (kernels/dec_half.hpp; OFF by default): the decoder's self-attention block on half Q | K | V.  (a) the default handle does not run
    it (decode_half_ok = 0, no reading); (b) switched on before finalize, the probe reads the form on unit-normal latents and keeps it exactly when the
    reading is <= MLDHIP_PROBE_TOL_HALF (the documented rule), on both synthetic weight families; (c) forced ("dec_half" 2), the reference's own ragged
    MldVae.decode fixture and the bs-64 pipeline fixture stay inside the feature / joint tolerances (|z| ~ 75: the easy case), and on unit-normal
    latents -- the case the form is opt-in for -- the joints against the oracle are printed and held to the 1e-3 contract on the first family;
    (d) the 96-row strip form gives the same features to rounding."""
    sdd = syn.make_denoiser_state_dict()
    mean, std = syn.make_mean_std()
    g = _gold(golden_dir, "pipeline_b64.npz")
    gd = _gold(golden_dir, "vae_decode_b3.npz")

    def mk(sdv, dh):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std)
        if dh:
            e.set_option("dec_half", dh)
        e.finalize()
        return e
    sdv1 = syn.make_vae_state_dict()
    e0 = mk(sdv1, 0)
    s0 = e0.numeric_status()
    assert s0["decode_half_ok"] == 0 and s0["probe_err_decode_half"] == -1.0 and s0["decode_split_ok"] == 1, s0
    e0.close()
    for fam, sdv in (("family1", sdv1), ("family2", syn.trained_like(sdv1))):
        e = mk(sdv, 1)
        s = e.numeric_status()
        print("dec_half probe", fam, {k: s[k] for k in ("decode_half_ok", "probe_err_decode_half", "probe_err_decode", "decode_split_ok")})
        assert s["probe_err_decode_half"] > s["probe_err_decode"] >= 0 and s["decode_half_ok"] == (1 if s["probe_err_decode_half"] <= _lib.PROBE_TOL_HALF else 0), s
        e.close()
    e = mk(sdv1, 2)
    assert e.numeric_status()["decode_half_ok"] == 1
    f3 = torch.full((3, 100, 263), float("nan"), device=dev)
    e.vae_decode(_cuda(gd["z"], dev), [int(x) for x in gd["lengths"]], f3)
    torch.cuda.synchronize()
    ef3 = np.abs(f3.cpu().numpy() - gd["feats"]).max()
    lat, feats, joints, _ = _run_sample(e, dev, syn.make_batch(64))
    ej = np.abs(joints.cpu().numpy() - g["joints"]).max()
    ops = O.TorchOps("float32")
    zu = syn._rng(41, "dec_half_unit").standard_normal((8, 1, 256)).astype(np.float32)
    lens = [196, 120, 196, 64, 33, 196, 150, 196]
    bv = O.to_backend(ops, sdv1)
    fr = O.vae_decode(ops, bv, ops.asarray(zu), lens)
    jr = ops.to_numpy(O.feats2joints(ops, fr, ops.asarray(mean), ops.asarray(std)))
    errs = {}
    for dh in (0, 2, 6):
        e.set_option("dec_half", dh)
        if dh == 6:
            e.finalize()                              # (a value the probe may veto: un-finalizes a probed handle)
        fu = torch.zeros(8, 196, 263, device=dev); ju = torch.zeros(8, 196, 22, 3, device=dev)
        e.vae_decode(_cuda(zu, dev), lens, fu)
        e.feats2joints(fu, 8, 196, ju)
        torch.cuda.synchronize()
        errs[dh] = (max(float(np.abs(ju.cpu().numpy()[i, :n] - jr[i, :n]).max()) for i, n in enumerate(lens)), fu.clone())
    print("dec_half forced: ragged decode fixture feats %.3e, bs-64 pipeline joints %.3e; unit-normal latents joints vs oracle: x3 %.3e, half %.3e"
          % (ef3, ej, errs[0][0], errs[2][0]))
    assert ef3 < 1e-4 and ej < 5e-4
    assert errs[0][0] < 5e-5 and errs[0][0] < errs[2][0] < 1e-3
    if e.numeric_status()["decode_half_ok"]:
        assert (errs[6][1] - errs[2][1]).abs().max().item() < 5e-5
    e.close()


def test_first_decoder_layer_projected_once_and_access_options_change_nothing(dev, golden_dir):
    """"dec_l0_once" (default on): decoder layer 0's in-projection over ONE sample's positional rows, read by every (sample, head)
    attention workgroup -- against the per-sample form: same kernels and products, so the reference's ragged MldVae.decode fixture
    (sample 0 is NOT the longest) and a 64-motion decode agree to the bit in exact fp32 and to fp32 rounding in the split mode (the
    shared projection takes the small-M GEMM shape there)."""
    gd = _gold(golden_dir, "vae_decode_b3.npz")
    b = syn.make_batch(64, "ragged", seed=77)
    lens64 = [int(x) for x in b.lengths]
    lens64[0] = 40                                 # sample 0 (whose rows carry the shared projection) is the shortest
    tm = max(lens64)
    for prec in (0, 1):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=prec)
        _load(e)
        outs = []
        for once in (1, 0):
            e.set_option("dec_l0_once", once)
            f3 = torch.full((3, 100, 263), float("nan"), device=dev)
            e.vae_decode(_cuda(gd["z"], dev), [int(x) for x in gd["lengths"]], f3)
            z = _cuda(syn._rng(5, "l0z").standard_normal((64, 1, 256)).astype(np.float32), dev)
            f64 = torch.full((64, tm, 263), float("nan"), device=dev)
            e.vae_decode(z, lens64, f64)
            torch.cuda.synchronize()
            assert np.abs(f3.cpu().numpy() - gd["feats"]).max() < 1e-4
            outs.append((f3.clone(), f64.clone()))
        d3, d64 = (outs[0][0] - outs[1][0]).abs().max().item(), (outs[0][1] - outs[1][1]).abs().max().item()
        print("dec_l0_once on vs off, precision %d: max diff %.3e (3 ragged) %.3e (64 motions)" % (prec, d3, d64))
        assert d3 < 2e-5 and d64 < 2e-5
        if prec == 1:
            # the key-blocked attention kernel (V through ds_read_b64_tr_b16) instead of the auto choice at 256 (sample, head) pairs
            e.set_option("flash_attn", 2)
            f = torch.full((64, tm, 263), float("nan"), device=dev)
            e.vae_decode(z, lens64, f)
            torch.cuda.synchronize()
            assert (f - outs[1][1]).abs().max().item() < 5e-5
        e.close()


def test_vae_encode_vs_reference_golden(eng, dev, golden_dir, oracle_weights):
    """Scope row 8f.1: MldVae.encode vs the reference's own Normal(mu, std) on the frozen inputs."""
    ops, _, bv = oracle_weights
    g = _gold(golden_dir, "vae_encode_b3.npz")
    lengths = [int(x) for x in g["lengths"]]
    eps = syn._rng(12, "eps").standard_normal((3, 256)).astype(np.float32)
    lat, mu, lv = (torch.empty(3, 256, device=dev) for _ in range(3))
    eng.vae_encode(_cuda(g["feats"], dev), lengths, 100, _cuda(eps, dev), lat, mu, lv)
    torch.cuda.synchronize()
    assert np.abs(mu.cpu().numpy() - g["mu"][:, 0]).max() < 1e-4
    assert np.abs(np.sqrt(np.exp(lv.cpu().numpy())) - g["std"][:, 0]).max() < 2e-4
    lr, _, _ = O.vae_encode(ops, bv, g["feats"], lengths, eps[:, None, :])
    assert np.abs(lat.cpu().numpy() - lr[:, 0]).max() < 5e-4
    # full-size batch: encode -> decode round trip stays finite and deterministic
    b = 64
    f = syn._rng(13, "f64").standard_normal((b, 196, 263)).astype(np.float32)
    z = torch.empty(b, 256, device=dev); m2 = torch.empty_like(z); l2 = torch.empty_like(z)
    eng.vae_encode(_cuda(f, dev), [196] * b, 196, None, None, m2, l2)
    eng.vae_encode(_cuda(f, dev), [196] * b, 196, None, None, z, l2)
    torch.cuda.synchronize()
    assert torch.isfinite(m2).all() and torch.equal(m2, z)


# ------------------------------------------------------------------ action-conditioned variant (BASELINE config 5)
ACTION_CFG = dict(condition=_lib.COND_ACTION, nclasses=12, vae_arch=_lib.VAE_ACTOR, vae_num_layers=6, num_layers=15, nfeats=150)


def _action_weights():
    dims = syn.ModelDims(num_layers=15, nfeats=150)
    return (syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), syn.make_actor_vae_state_dict())


@pytest.fixture(scope="module")
def aeng(dev):
    e = _lib.Engine(device=0, max_batch=256, max_frames=60, **ACTION_CFG)
    sdd, sdv = _action_weights()
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(sdv, "vae.")
    e.finalize()
    yield e
    e.close()


def test_action_denoiser_and_actor_decode_vs_golden(aeng, dev, golden_dir):
    """Single ops of the HumanAct12 variant against the reference modules' outputs (oracle/make_golden.py main_action)."""
    g = _gold(golden_dir, "action_ops_b4.npz")
    out = torch.empty(8, 1, 256, device=dev)
    aeng.denoiser_forward_action(_cuda(g["sample"], dev), 981, g["cond"].tolist(), out)
    lens = g["lengths"].tolist()
    feats = torch.full((4, max(lens), 150), 7.0, device=dev)
    aeng.vae_decode(_cuda(g["z"], dev), lens, feats)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - g["out_t981"]).max() < 1e-4
    f = feats.cpu().numpy()
    assert np.abs(f - g["feats"]).max() < 1e-4
    for i, n in enumerate(lens):
        assert np.all(f[i, n:] == 0)


def test_action_pipeline_b256_vs_golden(aeng, dev, golden_dir):
    """Config 5 at full size (B=256, T=60, 50 steps, guidance 7.5) against the reference-generated fixture; graph replay
    is deterministic and a second label set changes the result (the labels are read at replay time, not baked in)."""
    g = _gold(golden_dir, "action_b256.npz")
    acts, lat0, lens = syn.make_action_batch(256, nframes=60)
    lat = torch.empty(256, 1, 256, device=dev)
    feats = torch.empty(256, 60, 150, device=dev)
    x0 = _cuda(lat0, dev)
    aeng.sample_action(acts, x0, lens, lat, feats)
    torch.cuda.synchronize()
    l1, f1 = lat.cpu().numpy(), feats.cpu().numpy()
    assert np.abs(l1 - g["latents"]).max() < 5e-3          # |latents| ~ 70
    assert np.abs(f1[::8] - g["feats_every8"]).max() < 1e-3
    aeng.sample_action(acts, x0, lens, lat, feats)
    torch.cuda.synchronize()
    assert np.array_equal(f1, feats.cpu().numpy())
    aeng.sample_action((acts + 1) % 12, x0, lens, lat, feats)
    torch.cuda.synchronize()
    assert np.abs(feats.cpu().numpy() - f1).max() > 1e-2
    den, dec, _ = aeng.launch_counts()
    assert den == 2 + 50 * (15 * 4 + 7 + 1) and dec == 2 + 1 + 6 * 5 + 1


def test_action_mld_module_surface_on_gpu(dev):
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD

    cfg = C.load_config(os.path.join(C.CONFIG_DIR, "config_mld_humanact12.yaml"))
    E.configure("action", max_batch=8, max_frames=60)
    model = MLD(cfg, HipDataModule(cfg, nfeats=150, njoints=25, name="humanact12")).to(dev).eval()
    sdd, sdv = _action_weights()
    model.denoiser.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=True)
    model.vae.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()}, strict=True)
    acts, lat0, _ = syn.make_action_batch(5, 60)
    lengths = [60, 41, 60, 60, 17]
    rs = model.a2m_eval({"action": _cuda(acts.astype(np.int64), dev)[:, None], "length": lengths}, init_latents=_cuda(lat0, dev))
    ops = O.NumpyOps(np.float32)
    fr = O.sample_action(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), acts, lat0, lengths)
    assert rs["m_rst"].is_cuda and np.abs(rs["m_rst"].cpu().numpy() - fr).max() < 1e-3
    # several requests in ONE engine call (MLD.sample_many_action -> mldhip_sample_many) and the sampler's automatic requests per call:
    # 5 labels in chunks of 2 on an engine that holds 8 motions -> all three chunks coalesced, the same features per label
    outs = model.sample_many_action([(acts[:2], lengths[:2]), (acts[2:], lengths[2:])], init_latents=[_cuda(lat0[:2], dev), _cuda(lat0[2:], dev)])
    got = torch.cat([torch.nn.functional.pad(f, (0, 0, 0, 60 - f.shape[1])) for f, _ in outs]).cpu().numpy()
    for i, n in enumerate(lengths):
        assert np.abs(got[i, :n] - fr[i, :n]).max() < 1e-3
    from mld_hip.dp import DataParallelSampler
    smp = DataParallelSampler(model, batch_size=2, coalesce="auto")
    idx, feats = smp(actions=[int(a) for a in acts], lengths=lengths, init_latents=torch.from_numpy(lat0))
    assert idx == list(range(5)) and smp.last_coalesce == 3
    for i, n in enumerate(lengths):
        assert feats[i].shape == (n, 150) and np.abs(feats[i].numpy() - fr[i, :n]).max() < 1e-3
    E.drop_engines()


# ------------------------------------------------------------------ diffusion-only variant (BASELINE config 4)
NOVAE_CFG = dict(latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC, scheduler_type=_lib.SCHED_DDPM,
                 steps_offset=0)


def _novae_engine(steps, max_batch=64, max_frames=196):
    e = _lib.Engine(device=0, max_batch=max_batch, max_frames=max_frames, num_inference_steps=steps, **NOVAE_CFG)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    return e


@pytest.fixture(scope="module")
def neng(dev):
    e = _novae_engine(10)
    yield e
    e.close()


def test_novae_denoiser_vs_golden(neng, dev, golden_dir):
    """trans_dec denoiser on raw motion (d=512, heads of 128, 2-key cross-attention) vs the reference module's outputs:
    a ragged small batch at the first/last DDPM timestep and one call at config 4's full CFG shape (R=128, T=196)."""
    g = _gold(golden_dir, "novae_denoiser_b4.npz")
    lens = g["lengths"].tolist()
    for t in (999, 0):
        out = torch.full((4, 24, 263), 7.0, device=dev)
        neng.denoiser_forward_novae(_cuda(g["sample"], dev), t, _cuda(g["text_emb"], dev), lens, 24, out)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert np.abs(o - g[f"out_t{t}"]).max() < 1e-4
        assert np.all(o[1, 17:] == 0) and np.all(o[3, 9:] == 0)
    gf = _gold(golden_dir, "novae_denoiser_full.npz")
    b64 = syn.make_batch(64)
    xf = syn._rng(12, "nvfull").standard_normal((64, 196, 263)).astype(np.float32)
    out = torch.empty(128, 196, 263, device=dev)
    neng.denoiser_forward_novae(_cuda(np.concatenate([xf, xf]), dev), 500, _cuda(b64.text_emb, dev), gf["lengths"].tolist() * 2, 196, out)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy()[::16, ::7] - gf["out_t500_sub"]).max() < 1e-4


def test_novae_pipeline_vs_golden(neng, dev, golden_dir):
    """10 DDPM steps with CFG 7.5 and injected per-step noise, B=3 ragged, against the reference-module fixture.
    Tolerances: |feats| reaches 67 here and the fixture's own oracle-vs-reference floor is 1.6e-4 (feats) / 6.5e-4 (joints)."""
    g = _gold(golden_dir, "novae_pipeline_b3.npz")
    lens = g["lengths"].tolist()
    feats = torch.empty(3, 40, 263, device=dev)
    joints = torch.empty(3, 40, 22, 3, device=dev)
    neng.sample_novae(_cuda(g["text_emb"], dev), _cuda(g["init_latents"], dev), lens, _cuda(g["step_noise"], dev), 0, feats, joints)
    torch.cuda.synchronize()
    assert np.abs(feats.cpu().numpy() - g["feats"]).max() < 2e-3
    j = joints.cpu().numpy()
    for i, n in enumerate(lens):
        assert np.abs(j[i, :n] - g["joints"][i, :n]).max() < 3e-3
    assert neng.launch_counts()[0] == 3 + 10 * (1 + 2 + 9 * 7 + 2 + 1)      # (7 launches per layer since "cross_fold", round 6: 11 before; + the text tokens' fold in the prologue)


def test_novae_full_size_steps_vs_oracle_and_philox(dev):
    """Config 4's full shape (B=64, T=196): two DDPM steps against the oracle (torch-CPU backend), and the in-kernel Philox
    stream == the exposed stream == its numpy restatement."""
    e = _novae_engine(2)
    b = syn.make_batch(64)
    g = syn._rng(31, "nvfull_steps")
    lat0 = g.standard_normal((64, 196, 263)).astype(np.float32)
    n = lat0.size
    z = torch.empty(2, n, device=dev)
    for s in range(2):
        e.philox_normal(z[s], n, 1234, s)
    torch.cuda.synchronize()
    zr = O.philox_normal(n, 1234, 1)
    assert np.abs(z[1].cpu().numpy() - zr).max() < 5e-5
    assert abs(float(z.mean())) < 1e-3 and abs(float(z.std()) - 1.0) < 1e-3
    f1 = torch.empty(64, 196, 263, device=dev)
    f2 = torch.empty_like(f1)
    text, x0 = _cuda(b.text_emb, dev), _cuda(lat0, dev)
    e.sample_novae(text, x0, b.lengths, None, 1234, f1, None)                        # in-kernel Philox
    e.sample_novae(text, x0, b.lengths, z.view(2, 64, 196, 263), 0, f2, None)        # the same draws, injected
    torch.cuda.synchronize()
    assert torch.equal(f1, f2)
    ops = O.TorchOps()
    fr = O.sample_novae(ops, O.to_backend(ops, syn.make_novae_denoiser_state_dict()), ops.asarray(b.text_emb), ops.asarray(lat0),
                        b.lengths, ops.asarray(z.view(2, 64, 196, 263).cpu().numpy()), steps=2)
    assert np.abs(f1.cpu().numpy() - ops.to_numpy(fr)).max() < 1e-3
    e.close()


def test_novae_mld_module_surface_on_gpu(dev, golden_dir):
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder

    cfg = C.load_config(os.path.join(C.CONFIG_DIR, "config_novae_humanml3d.yaml"), overrides={"model.scheduler.num_inference_timesteps": 10})
    E.configure("novae", max_batch=4, max_frames=64)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev).eval()
    model.denoiser.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_novae_denoiser_state_dict().items()}, strict=True)
    assert model.fused and model.vae is None
    g = _gold(golden_dir, "novae_pipeline_b3.npz")
    lens = g["lengths"].tolist()
    joints, feats = model.sample_novae(_cuda(g["text_emb"], dev), lens, _cuda(g["init_latents"], dev), _cuda(g["step_noise"], dev))
    assert np.abs(feats.cpu().numpy() - g["feats"]).max() < 2e-3
    z = model._diffusion_reverse(_cuda(g["text_emb"], dev), lens, _cuda(g["init_latents"], dev), _cuda(g["step_noise"], dev))
    assert np.abs(z.permute(1, 0, 2).cpu().numpy() - g["feats"]).max() < 2e-3
    E.drop_engines()


def test_novae_data_parallel_sampler_with_batches_in_flight(dev):
    """DataParallelSampler(in_flight=2) on the diffusion-only variant: chunks on two rotating streams (two workspaces of one handle) give
    the SAME motions as the serial loop -- start noise and the DDPM scheduler's per-step draws are pinned per prompt, chunks share one
    Tmax (the trans_dec denoiser attends over the padded batch)."""
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.dp import DataParallelSampler
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder

    E.drop_engines()
    steps = 5                              # (a divisor of the 1000 training steps)
    cfg = C.load_config(os.path.join(C.CONFIG_DIR, "config_novae_humanml3d.yaml"), overrides={"model.scheduler.num_inference_timesteps": steps})
    E.configure("novae", max_batch=4, max_frames=48, max_in_flight=2)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev).eval()
    model.denoiser.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_novae_denoiser_state_dict().items()}, strict=True)
    n = 14
    texts = [f"prompt number {i}" for i in range(n)]
    lengths = [48 if i % 4 == 0 else 20 + 3 * (i % 7) for i in range(n)]          # every chunk of 4 holds a 48-frame motion: equal Tmax
    g = torch.Generator().manual_seed(11)
    lat0 = torch.randn(n, 48, 263, generator=g)
    sn = torch.randn(steps, n, 48, 263, generator=g)
    idx1, serial = DataParallelSampler(model, batch_size=4, in_flight=1)(texts, lengths, init_latents=lat0, step_noise=sn)
    idx2, flight = DataParallelSampler(model, batch_size=4, in_flight=2)(texts, lengths, init_latents=lat0, step_noise=sn)
    assert idx1 == idx2 == list(range(n)) and len(serial) == len(flight) == n
    for a, b, ln in zip(serial, flight, lengths):
        assert a.shape == b.shape == (ln, 22, 3) and bool(torch.isfinite(b).all()) and torch.equal(a, b)
    E.configure("novae", max_batch=64, max_frames=196, max_in_flight=1)
    E.drop_engines()


def test_graph_replay_is_independent_of_caller_buffers(eng, dev):
    """Fresh input/output tensors on every call (what MLD.forward does) must replay the same captured graph correctly."""
    b = syn.make_batch(5, [60, 33, 60, 41, 8])
    outs = []
    for rep in range(3):
        text, x0 = _cuda(b.text_emb, dev).clone(), _cuda(b.init_latents, dev).clone()
        pad = torch.empty(1000 * (rep + 1), device=dev)             # shift the allocator so the addresses differ
        joints = torch.full((5, 60, 22, 3), float("nan"), device=dev)
        lat = torch.full((5, 1, 256), float("nan"), device=dev)
        eng.sample(text, x0, b.lengths, lat, None, joints)
        torch.cuda.synchronize()
        outs.append((joints.cpu().numpy(), lat.cpu().numpy(), joints.data_ptr()))
        del pad
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


def test_batches_in_flight_on_three_streams_are_exact(eng, dev):
    """max_in_flight = 3: six different batches issued round-robin on three HIP streams (overlapping on the GPU) must give
    bit-identical motions to the same batches sampled one at a time on the single-workspace engine."""
    e3 = _lib.Engine(device=0, max_batch=64, max_frames=196, max_in_flight=3)
    _load(e3)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    jobs = []
    for i in range(6):
        b = syn.make_batch(64 if i % 2 == 0 else 17, None, seed=100 + i, max_len=196 if i < 4 else 120)
        jobs.append((b, _cuda(b.text_emb, dev), _cuda(b.init_latents, dev),
                     torch.empty(len(b.lengths), max(b.lengths), 22, 3, device=dev)))
    torch.cuda.synchronize()
    for rep in range(2):                                   # second round replays the captured graphs
        for i, (b, text, x0, joints) in enumerate(jobs):
            e3.sample(text, x0, b.lengths, None, None, joints, streams[i % 3].cuda_stream)
    torch.cuda.synchronize()
    for b, text, x0, joints in jobs:
        ref = torch.empty_like(joints)
        eng.sample(text, x0, b.lengths, None, None, ref)
        torch.cuda.synchronize()
        assert torch.equal(joints, ref)
    e3.close()


def test_data_parallel_sampler_with_batches_in_flight(dev):
    """DataParallelSampler(in_flight=3) on one rank: chunks on rotating streams give the same motions as the serial loop
    (noise comes from torch.randn inside MLD.sample, so both runs are seeded identically per chunk order)."""
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.dp import DataParallelSampler
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder

    E.drop_engines()
    cfg = C.load_config()
    E.configure("text", max_batch=8, max_frames=196, max_in_flight=3)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev).eval()
    texts = [f"prompt number {i}" for i in range(20)]
    lengths = [40 + 7 * (i % 9) for i in range(20)]

    orig = model.sample          # wrap it: the same start noise whichever stream / order a chunk runs in

    def seeded(text_emb, lens, init_latents=None):
        g = torch.Generator(device=dev).manual_seed(sum(lens))
        return orig(text_emb, lens, torch.randn((len(lens), 1, 256), device=dev, generator=g))
    model.sample = seeded
    idx1, serial = DataParallelSampler(model, batch_size=8, in_flight=1)(texts, lengths)
    idx3, flight = DataParallelSampler(model, batch_size=8, in_flight=3)(texts, lengths)
    assert idx1 == idx3 == list(range(20))
    for a, b, n in zip(serial, flight, lengths):
        assert a.shape == (n, 22, 3) and torch.equal(a, b)
    E.configure("text", max_in_flight=1)
    E.drop_engines()


def test_extreme_shapes_vs_oracle(dev, oracle_weights):
    """Edges of the capacity envelope: the longest supported motion (288 frames: 18 key tiles in the attention kernel)
    next to a 1-frame motion, and a batch of one 1-frame motion."""
    ops, bd, bv = oracle_weights
    mean, std = syn.make_mean_std()
    e = _lib.Engine(device=0, max_batch=2, max_frames=288, num_inference_steps=4)
    _load(e)
    for lens in ([288, 1], [1]):
        b = syn.make_batch(len(lens), lens, seed=55)
        B, T = len(lens), max(lens)
        joints = torch.empty(B, T, 22, 3, device=dev)
        feats = torch.empty(B, T, 263, device=dev)
        e.sample(_cuda(b.text_emb, dev), _cuda(b.init_latents, dev), lens, None, feats, joints)
        torch.cuda.synchronize()
        jr, fr, _ = O.sample(ops, bd, bv, b.text_emb, b.init_latents, lens, mean, std, steps=4, return_intermediates=True)
        assert np.abs(feats.cpu().numpy() - fr).max() < 1e-4
        j = joints.cpu().numpy()
        for i, n in enumerate(lens):
            assert np.abs(j[i, :n] - jr[i, :n]).max() < 1e-3
        assert np.all(feats.cpu().numpy()[-1, lens[-1]:] == 0)
    with pytest.raises(_lib.MldHipError):
        _lib.Engine(device=0, max_frames=289)
    e.close()


def test_actor_encode_vs_golden(aeng, dev, golden_dir):
    """ActorVae.encode (actor_vae.py:64-76,121-175) vs the reference module's Normal(mu, std) on a ragged batch."""
    g = _gold(golden_dir, "actor_encode_b3.npz")
    lens = g["lengths"].tolist()
    mu = torch.empty(3, 256, device=dev)
    lv = torch.empty_like(mu)
    aeng.vae_encode(_cuda(g["feats"], dev), lens, 60, None, None, mu, lv)
    torch.cuda.synchronize()
    assert np.abs(mu.cpu().numpy() - g["mu"]).max() < 1e-4
    assert np.abs(np.sqrt(np.exp(lv.cpu().numpy())) - g["std"]).max() < 1e-4


def test_one_workspace_two_streams_never_share_it(dev, eng):
    """with max_in_flight = 1 two calls on different streams must be ordered behind each other on the device
    (one activation workspace).  Interleave many calls on two non-default streams and compare with serial results."""
    e = _lib.Engine(device=0, max_batch=8, max_frames=64, max_in_flight=1, num_inference_steps=10)
    _load(e)
    b1, b2 = syn.make_batch(8, [64] * 8, seed=5), syn.make_batch(8, [40, 64, 33, 64, 12, 64, 64, 50], seed=6)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ins = [(_cuda(b.text_emb, dev), _cuda(b.init_latents, dev), b.lengths) for b in (b1, b2)]
    ref = []
    for text, lat0, lens in ins:                                # serial reference on the default stream
        j = torch.empty(8, 64, 22, 3, device=dev)
        e.sample(text, lat0, lens, None, None, j)
        torch.cuda.synchronize()
        ref.append(j.clone())
    torch.cuda.synchronize()
    outs = [[torch.empty(8, 64, 22, 3, device=dev) for _ in range(6)] for _ in range(2)]
    for it in range(6):                                         # no host sync in between: the streams race unless the engine orders them
        for k, st in enumerate((s1, s2)):
            text, lat0, lens = ins[k]
            e.sample(text, lat0, lens, None, None, outs[k][it], st.cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        for it in range(6):
            assert torch.equal(outs[k][it], ref[k]), f"stream {k} iteration {it} differs: workspace shared by two calls"
    e.close()


def test_rccl_world_size_2_data_parallel(tmp_path):
    """The same job as test_nccl_broadcast_and_dp_sampler at world size 2 exactly: two ranks on two devices, one RCCL broadcast over xGMI, each rank's motions
    compared with a single-process run.  SKIPPED (not passed) where fewer than two MI355X are visible -- every lease of rounds 1-5 had one."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices; this box has %d" % torch.cuda.device_count())
    _nccl_job(tmp_path, 2, "29533")


def test_nccl_broadcast_and_dp_sampler(tmp_path):
    """RCCL path end to end at world_size = number of visible GPUs (1 on the test box, 8 on a full node): the NCCL-backend
    dist.broadcast of the packed weights is EXECUTED at every world size, each rank samples its shard on its own GPU."""
    _nccl_job(tmp_path, torch.cuda.device_count(), "29531")


def _nccl_job(tmp_path, world, port):
    import json
    import subprocess
    import sys
    out = str(tmp_path / "nccl.json")
    here = os.path.dirname(os.path.abspath(__file__))
    nprompts = 6 * world + 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(here, "dp_worker_nccl.py"), out, str(nprompts)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = json.load(open(out))
    assert got["backend"] == "nccl" and got["world"] == world
    assert len({tuple(x) for x in got["ranks"]}) == world and len({x[2] for x in got["ranks"]}) == world    # distinct devices
    assert sorted(i for idx in got["indices"] for i in idx) == list(range(nprompts))                         # every prompt exactly once
    assert all(got["ok"])
    # ... and every rank's motions equal a single-process run of the same prompts with the same per-prompt noise
    from mld_hip import engine as E
    from mld_hip.config import load_config
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder
    E.drop_engines()
    E.configure("text", max_batch=4, max_frames=64, max_in_flight=1)
    dev = torch.device("cuda:0")
    cfg = load_config()
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev)
    sd = {**{"denoiser." + k: torch.from_numpy(v) for k, v in syn.make_denoiser_state_dict().items()},
          **{"vae." + k: torch.from_numpy(v) for k, v in syn.make_vae_state_dict().items()}}
    model.load_state_dict(sd, strict=False)
    texts = ["prompt %d" % i for i in range(nprompts)]
    lengths = [24 + 8 * (i % 5) for i in range(nprompts)]
    lat0 = _cuda(syn._rng(4242, "nccl_dp").standard_normal((nprompts, 1, 256)).astype(np.float32), dev)
    mot = np.load(out + ".npz")
    for s0 in range(0, nprompts, 4):
        ref = model({"text": texts[s0:s0 + 4], "length": lengths[s0:s0 + 4]}, init_latents=lat0[s0:s0 + 4])
        for k, j in enumerate(ref):
            assert np.abs(mot[f"m_{s0 + k}"] - j.numpy()).max() < 1e-4, (s0 + k)
    E.drop_engines()


def test_mld_forward_through_clip_adapter_on_gpu(dev, tmp_path, oracle_weights):
    """MLD.forward with the real CLIP adapter class (random-init CLIPModel directory, see test_text_encoder.py) on the GPU:
    text -> CLIP on PyTorch-ROCm -> [2B,1,768] -> libmldhip sample() -> joints, vs the oracle fed the same embeddings."""
    from test_text_encoder import make_clip_dir
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import MldTextEncoder

    E.drop_engines()
    d, _ = make_clip_dir(tmp_path)
    cfg = C.load_config()
    E.configure("text", max_batch=4, max_frames=64)
    enc = MldTextEncoder(d)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=enc).to(dev).eval()
    assert model.fused and next(enc.text_model.parameters()).is_cuda
    texts, lengths = ["a man walks.", "a person runs.", ""], [40, 33, 64]
    lat0 = _cuda(syn.make_batch(3, lengths).init_latents, dev)
    joints = model({"text": texts, "length": lengths}, init_latents=lat0)
    emb = enc([""] * 3 + texts)
    assert emb.is_cuda and tuple(emb.shape) == (6, 1, 768)
    ops, bd, bv = oracle_weights
    mean, std = syn.make_mean_std()
    jr = O.sample(ops, bd, bv, emb.cpu().numpy(), lat0.cpu().numpy(), lengths, mean, std)
    for i, n in enumerate(lengths):
        assert tuple(joints[i].shape) == (n, 22, 3) and np.abs(joints[i].numpy() - jr[i, :n]).max() < 1e-3
    E.drop_engines()


def test_sample_many_coalesced_requests_vs_reference_golden(dev, golden_dir, oracle_weights):
    """mldhip_sample_many: four requests as ONE chain (256 motions -> 1 536 token rows: the throughput kernels of
    kernels/strip.hpp are picked automatically).  Request 0 is the reference-generated B=64 / T=196 fixture; the others are
    ragged (own Tmax each, scattered with their own row pitch) and must equal their own mldhip_sample calls within the
    re-association noise of the two kernel families."""
    big = _lib.Engine(device=0, max_batch=160, max_frames=196)
    _load(big)
    small = _lib.Engine(device=0, max_batch=64, max_frames=196)
    _load(small)
    g = _gold(golden_dir, "pipeline_b64.npz")
    batches = [syn.make_batch(64), syn.make_batch(40, "ragged", seed=3), syn.make_batch(7, [33, 1, 196, 64, 65, 12, 100], seed=4),
               syn.make_batch(49, None, seed=5, max_len=120)]
    reqs = []
    for b in batches:
        B, T = len(b.lengths), max(b.lengths)
        reqs.append(dict(text_emb=_cuda(b.text_emb, dev), init_latents=_cuda(b.init_latents, dev), lengths=b.lengths,
                         latents_out=torch.empty(B, 1, 256, device=dev), feats_out=torch.empty(B, T, 263, device=dev),
                         joints_out=torch.empty(B, T, 22, 3, device=dev)))
    for _ in range(2):                                  # second call replays the captured graph of (160, 196)
        big.sample_many(reqs)
    torch.cuda.synchronize()
    q = reqs[0]
    assert np.abs(q["latents_out"].cpu().numpy() - g["latents"]).max() < 5e-3
    assert np.abs(q["feats_out"].cpu().numpy()[:, -1] - g["feats_frame_last"]).max() < 2e-4
    assert np.abs(q["joints_out"].cpu().numpy() - g["joints"]).max() < 1e-3
    for b, q in zip(batches[1:], reqs[1:]):
        B, T = len(b.lengths), max(b.lengths)
        lat, feats, joints = torch.empty(B, 1, 256, device=dev), torch.empty(B, T, 263, device=dev), torch.empty(B, T, 22, 3, device=dev)
        small.sample(q["text_emb"], q["init_latents"], b.lengths, lat, feats, joints)
        torch.cuda.synchronize()
        assert (q["latents_out"] - lat).abs().max().item() < 5e-3
        assert (q["joints_out"] - joints).abs().max().item() < 1e-3 and (q["feats_out"] - feats).abs().max().item() < 1e-3
        for i, n in enumerate(b.lengths):
            assert torch.all(q["feats_out"][i, n:] == 0)
    with pytest.raises(_lib.MldHipError):
        big.sample_many(reqs + reqs[:1])                # 224 motions > max_batch
    big.close()
    small.close()


def test_throughput_kernels_single_call_vs_oracle(dev, oracle_weights):
    """The two loop-kernel families on the same 8-motion batch vs the oracle (loop_kernel option, graphs dropped in between)."""
    ops, bd, bv = oracle_weights
    mean, std = syn.make_mean_std()
    e = _lib.Engine(device=0, max_batch=8, max_frames=64, num_inference_steps=10)
    _load(e)
    b = syn.make_batch(8, [64, 40, 33, 64, 12, 1, 64, 50], seed=8)
    jr, fr, lr = O.sample(ops, bd, bv, b.text_emb, b.init_latents, b.lengths, mean, std, steps=10, return_intermediates=True)
    outs = []
    for fam in (1, 2):
        e.set_option("loop_kernel", fam)
        lat, joints = torch.empty(8, 1, 256, device=dev), torch.empty(8, 64, 22, 3, device=dev)
        for _ in range(2):
            e.sample(_cuda(b.text_emb, dev), _cuda(b.init_latents, dev), b.lengths, lat, None, joints)
        torch.cuda.synchronize()
        assert np.abs(lat.cpu().numpy() - lr).max() < 2e-3 and np.abs(joints.cpu().numpy() - jr).max() < 1e-3
        outs.append(joints.clone())
    assert not torch.equal(outs[0], outs[1])            # the option really switched kernels (different summation order)
    e.close()


def test_novae_full_length_1000_steps_vs_reference_golden(dev, golden_dir):
    """BASELINE config 4 at its real length (configs/modules_novae/scheduler.yaml:16-29: 1000 DDPM steps; loop mld.py:323-346):
    B = 2, lengths [196, 150], noise from the in-kernel Philox stream (seed, step) -- the same stream the fixture's generator
    regenerated (oracle/make_golden_novae1000.py; reference MldDenoiser in the loop, fp32).  The fixture stores how far a
    float64 evaluation drifts from the reference's float32 one on this 1000-step map (features |x| ~ 148: 4e-4; joints, which
    integrate 196 frames of yaw / root velocity on top: 2e-2); the engine must stay within a small multiple of that floor."""
    g = _gold(golden_dir, "novae_pipeline_1000.npz")
    lens = [int(x) for x in g["lengths"]]
    B, T = len(lens), max(lens)
    e = _lib.Engine(device=0, max_batch=B, max_frames=T, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=int(g["steps"]), steps_offset=0)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    b = syn.make_batch(B, lens, seed=int(g["batch_seed"]))
    lat0 = syn._rng(int(g["lat0_seed"]), "nv1000").standard_normal((B, T, 263)).astype(np.float32)
    feats, joints = torch.empty(B, T, 263, device=dev), torch.empty(B, T, 22, 3, device=dev)
    e.sample_novae(_cuda(b.text_emb, dev), _cuda(lat0, dev), lens, None, int(g["seed"]), feats, joints)
    torch.cuda.synchronize()
    f, j = feats.cpu().numpy(), joints.cpu().numpy()
    ef = max(np.abs(f[i, :n] - g["feats"][i, :n]).max() for i, n in enumerate(lens))
    ej = max(np.abs(j[i, :n] - g["joints"][i, :n]).max() for i, n in enumerate(lens))
    print("novae 1000 steps: feats err %.3e (f64 floor %.3e, |x| max %.1f)  joints err %.3e (floor %.3e)"
          % (ef, float(g["f64_diff_feats"]), float(g["feats_absmax"]), ej, float(g["f64_diff_joints"])))
    assert np.isfinite(f).all() and np.isfinite(j).all()
    assert ef < 10 * float(g["f64_diff_feats"]) and ej < 10 * float(g["f64_diff_joints"])
    e.close()


def test_ragged_bs64_uniform_length_mix_vs_oracle(eng, dev):
    """BASELINE config 2 batch size with the realistic length mix of SURVEY.md §8(d) (uniform in {40..196 step 4}, seed 1234,
    one motion at 196): the decoder skips every all-padding row tile (gemm.hpp) and key tile (attention.hpp) -- exactness of
    that shortcut at full size, against the oracle (torch-CPU backend: same arithmetic as the numpy one, multi-threaded)."""
    rng = np.random.Generator(np.random.PCG64(1234))
    lens = [int(v) for v in rng.choice(np.arange(40, 197, 4), 64)]
    lens[0] = 196
    b = syn.make_batch(64, lens, seed=1234)
    lat, feats, joints = torch.empty(64, 1, 256, device=dev), torch.empty(64, 196, 263, device=dev), torch.empty(64, 196, 22, 3, device=dev)
    eng.sample(_cuda(b.text_emb, dev), _cuda(b.init_latents, dev), lens, lat, feats, joints)
    torch.cuda.synchronize()
    ops = O.TorchOps("float32")
    mean, std = syn.make_mean_std()
    jr, fr, lr = O.sample(ops, O.to_backend(ops, syn.make_denoiser_state_dict()), O.to_backend(ops, syn.make_vae_state_dict()),
                          ops.asarray(b.text_emb), ops.asarray(b.init_latents), lens, ops.asarray(mean), ops.asarray(std),
                          return_intermediates=True)
    jr, fr, lr = (ops.to_numpy(x) for x in (jr, fr, lr))
    f, j = feats.cpu().numpy(), joints.cpu().numpy()
    assert np.abs(lat.cpu().numpy() - lr).max() < 5e-3
    ej = max(np.abs(j[i, :n] - jr[i, :n]).max() for i, n in enumerate(lens))
    ef = max(np.abs(f[i, :n] - fr[i, :n]).max() for i, n in enumerate(lens))
    print("ragged bs64: feats err %.3e joints err %.3e (mean length %.1f)" % (ef, ej, float(np.mean(lens))))
    assert ef < 1e-3 and ej < 1e-3
    for i, n in enumerate(lens):
        assert np.all(f[i, n:] == 0)


def test_reduced_precision_modes_run_and_stay_sane(dev, golden_dir):
    """MLDHIP_PREC_BF16 is a REPORTED mode (bench.py prints its error): here only that it runs on the hardware MFMA form, produces
    finite motions of the right shape, and sits where the format puts it -- within a few percent of the reference latents (|x| ~ 80);
    that precision 3 (the fp8 denoiser mode of ABI <= 4, retired in round 6: it met no tolerance and was slower than split-f16) is refused
    at mldhip_create with a message that says so; and that the split-bf16 mode of the
    diffusion-only variant stays within 5e-3 of the reference on the 10-step fixture (|x| ~ 67; fp32: 2e-4)."""
    g = _gold(golden_dir, "pipeline_b64.npz")
    b = syn.make_batch(64)
    errs = {}
    with pytest.raises(_lib.MldHipError, match="retired"):
        _lib.Engine(device=0, max_batch=64, max_frames=196, precision=3)
    for prec in (2,):
        e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=prec)
        _load(e)
        lat, _, joints, _ = _run_sample(e, dev, b)
        assert torch.isfinite(joints).all() and torch.isfinite(lat).all()
        errs[prec] = float(np.abs(lat.cpu().numpy() - g["latents"]).max())
        e.close()
    print("latent error vs reference: bf16 %.3f (|x| ~ 80)" % errs[2])
    assert 1e-3 < errs[2] < 5.0
    gn = _gold(golden_dir, "novae_pipeline_b3.npz")
    lens = [int(x) for x in gn["lengths"]]
    e = _lib.Engine(device=0, max_batch=3, max_frames=40, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                    scheduler_type=_lib.SCHED_DDPM, num_inference_steps=10, steps_offset=0, precision=1)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    ns = e.numeric_status()                             # the range probe covers this variant too (one denoiser call, split vs fp32 kernels)
    assert ns["probed"] == 1 and ns["decode_split_ok"] == 1 and 0 <= ns["probe_err_decode"] <= _lib.PROBE_TOL, ns
    e.set_option("gemm_small_m", 0)                     # 240 rows: drive the staged (precision-aware) GEMMs as at full size
    feats = torch.empty(3, 40, 263, device=dev)
    e.sample_novae(_cuda(gn["text_emb"], dev), _cuda(gn["init_latents"], dev), lens, _cuda(gn["step_noise"], dev), 0, feats, None)
    torch.cuda.synchronize()
    ef = max(np.abs(feats.cpu().numpy()[i, :n] - gn["feats"][i, :n]).max() for i, n in enumerate(lens))
    print("diffusion-only, split-bf16 GEMMs, 10 steps: feats err %.3e" % ef)
    assert 2e-5 < ef < 5e-3
    e.close()


def test_novae_split_f16_full_shape_kernels_vs_reference_golden(dev, golden_dir):
    """precision = F16X3 at config 4's full CFG shape (R = 128, T = 196: M = 25 088 rows, 512 (sample, head) pairs) -- the shape at which the
    software-pipelined 128 x 256 GEMM tile ("gemm_pipe", kernels/gemm_pipe.hpp) and the head-dim-128 key-blocked attention ("flash_attn",
    attn_flash128_x3_kernel) are what runs: one denoiser call against the reference module's fixture for every combination; the two GEMM
    tiles take the same products in the same order (identical to the bit, three calls each: a race between workgroup phases would show
    here), the two attention kernels sum in another order (close, not equal)."""
    gf = _gold(golden_dir, "novae_denoiser_full.npz")
    b64 = syn.make_batch(64)
    xf = syn._rng(12, "nvfull").standard_normal((64, 196, 263)).astype(np.float32)
    x, text, lens = _cuda(np.concatenate([xf, xf]), dev), _cuda(b64.text_emb, dev), gf["lengths"].tolist() * 2
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, num_inference_steps=10, precision=1, **NOVAE_CFG)
    e.load_state_dict(syn.make_novae_denoiser_state_dict(), "denoiser.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    assert e.numeric_status()["decode_split_ok"] == 1
    outs = {}
    for pipe, flash in ((1, 1), (0, 1), (1, 0), (0, 0)):
        e.set_option("gemm_pipe", pipe)
        e.set_option("flash_attn", flash)
        runs = []
        for _ in range(3):
            out = torch.empty(128, 196, 263, device=dev)
            e.denoiser_forward_novae(x, 500, text, lens, 196, out)
            torch.cuda.synchronize()
            runs.append(out)
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), (pipe, flash)
        outs[(pipe, flash)] = runs[0]
        err = float(np.abs(runs[0].cpu().numpy()[::16, ::7] - gf["out_t500_sub"]).max())
        print("diffusion-only denoiser, split-f16, gemm_pipe %d flash_attn %d: err vs the reference fixture %.3e" % (pipe, flash, err))
        assert err < 3e-4
    assert torch.equal(outs[(1, 1)], outs[(0, 1)]) and torch.equal(outs[(1, 0)], outs[(0, 0)])
    d = float((outs[(1, 1)] - outs[(1, 0)]).abs().max())
    assert 0 < d < 1e-4, d
    e.close()


def test_key_blocked_attention_matches_whole_kv_attention_on_gpu(dev, golden_dir):
    """precision = BF16X3_DECODE: attn_flash_x3_kernel ("flash_attn" = 2; auto picks it from 512 (sample, head) pairs up) vs
    attn_decode_x3_kernel (= 0) on the benchmarked shape, full-length and ragged: joints within 5e-5 of each other (summation
    order, unnormalised P operand), and the key-blocked path inside the joints contract on the reference fixture."""
    g = _gold(golden_dir, "pipeline_b64.npz")
    e = _lib.Engine(device=0, max_batch=64, max_frames=196, precision=1)
    _load(e)
    outs = {}
    for name, b in (("full", syn.make_batch(64)), ("ragged", syn.make_batch(64, "ragged", seed=5))):
        for fl in (2, 0):
            e.set_option("flash_attn", fl)
            lat, feats, joints, _ = _run_sample(e, dev, b)
            assert torch.isfinite(joints).all()
            outs[name, fl] = (feats.clone(), joints.clone())
        df = float((outs[name, 2][0] - outs[name, 0][0]).abs().max())
        dj = float((outs[name, 2][1] - outs[name, 0][1]).abs().max())
        print("%s: key-blocked vs whole-K/V attention: feats %.2e joints %.2e" % (name, df, dj))
        assert 0 < df < 5e-5 and dj < 1e-4
        for i, n in enumerate(b.lengths):
            assert torch.all(outs[name, 2][0][i, n:] == 0)
    err = float(np.abs(outs["full", 2][1].cpu().numpy() - g["joints"]).max())
    print("key-blocked attention, joints vs reference golden: %.2e" % err)
    assert err < 1e-3
    e.close()


def test_config3_shape_512_prompts_through_the_dp_sampler(dev):
    """BASELINE config 3 on the ranks this box has (1): 512 synthetic prompts, bs 64, sharded by DataParallelSampler -- every
    prompt exactly once and in order, and the coalesced + overlapped serving shape (4 chunks per engine call, 2 calls in flight)
    returns the same motions as one chunk per call (start noise pinned per chunk)."""
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.dp import DataParallelSampler
    from mld_hip.mld import MLD
    from mld_hip.text_encoder import SyntheticTextEncoder

    E.drop_engines()
    cfg = C.load_config()
    E.configure("text", max_batch=256, max_frames=196, max_in_flight=2)
    model = MLD(cfg, HipDataModule(cfg), text_encoder=SyntheticTextEncoder()).to(dev).eval()
    n = 512
    texts = [f"synthetic prompt {i}" for i in range(n)]
    rng = np.random.Generator(np.random.PCG64(3))
    lengths = [int(v) for v in rng.choice(np.arange(40, 197, 4), n)]

    def noise(lens):                       # the same start noise for a chunk whichever call shape serves it
        g = torch.Generator(device=dev).manual_seed(1000003 * lens[0] + sum(lens))
        return torch.randn((len(lens), 1, 256), device=dev, generator=g)
    orig_sample, orig_many = model.sample, model.sample_many
    model.sample = lambda emb, lens, init_latents=None: orig_sample(emb, lens, noise(lens))
    model.sample_many = lambda reqs, init_latents=None: orig_many(reqs, [noise(l) for _, l in reqs])
    idx1, one = DataParallelSampler(model, batch_size=64, in_flight=1, coalesce=1)(texts, lengths)
    idx4, many = DataParallelSampler(model, batch_size=64, in_flight=2, coalesce=4)(texts, lengths)
    auto = DataParallelSampler(model, batch_size=64, coalesce="auto")             # picked from the shard (8 chunks) and the engine (max_batch 256)
    idxa, manya = auto(texts, lengths)
    assert auto.last_coalesce == 4
    assert idx1 == idx4 == idxa == list(range(n)) and len(one) == len(many) == len(manya) == n
    worst = 0.0
    for a, b, c_, ln in zip(one, many, manya, lengths):
        assert a.shape == b.shape == c_.shape == (ln, 22, 3) and bool(torch.isfinite(b).all())
        worst = max(worst, float((a - b).abs().max()), float((a - c_).abs().max()))
    print("512 prompts: coalesced x in-flight vs one chunk per call, max-abs joints difference %.3e" % worst)
    assert worst < 1e-3                    # latency vs throughput kernel family: summation order only
    E.configure("text", max_batch=64, max_in_flight=1)
    E.drop_engines()


@pytest.mark.parametrize("NREQ", [20, 32])
def test_headline_serving_shape_every_motion_within_tolerance(dev, golden_dir, NREQ):
    """The shape bench.py's headline measures, all of it: split precision mode (precision = 1: split-f16 MFMAs in the reverse loop and
    the decoder), ONE mldhip_sample_many call of NREQ bs-64 requests -- 20 = 1 280 motions is what the driver's `bench.py --steps 20`
    issues (160 workgroups of the persistent loop on 256 CUs), 32 = 2 048 motions fills the chip -- T = 196: the reverse loop runs as the
    sample-major persistent launch (kernels/loop_fused.hpp), the decoder on the throughput shapes (key-blocked attention, fused FFN).
    Checked against (a) the reference's own output for request 0 (the pipeline_b64 fixture: reference MldDenoiser / MldVae /
    recover_from_ric, mld.py:290-360, mld_vae.py:186-248), (b) the CPU oracle for two more requests, (c) for EVERY one of the 2 048
    motions the exact-fp32 engine on the latency kernels (itself held to the fixture at 1e-4-class errors by the tests above), with a
    bound that leaves room for that engine's own distance to the reference; then the same call on the uniform {40..196} length mix.
    Tolerance: 1e-3 max-abs on the joints (north star)."""
    ops = O.TorchOps("float32")
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    bd, bv = O.to_backend(ops, sdd), O.to_backend(ops, sdv)
    mean, std = syn.make_mean_std()
    g = _gold(golden_dir, "pipeline_b64.npz")
    big = _lib.Engine(device=0, max_batch=64 * NREQ, max_frames=196, precision=1)   # MLDHIP_PREC_F16X3 (split 16-bit arithmetic)
    _load(big)
    ns = big.numeric_status()              # the range probe of finalize kept both stages on the split kernels (mldhip.h "Range contract")
    assert ns["probed"] == 1 and ns["loop_split_ok"] == 1 and ns["decode_split_ok"] == 1, ns
    assert 0 <= ns["probe_err_loop"] <= _lib.PROBE_TOL and 0 <= ns["probe_err_decode"] <= _lib.PROBE_TOL, ns
    exact = _lib.Engine(device=0, max_batch=64, max_frames=196)
    _load(exact)
    report = {}
    for mix in ("full", "ragged"):
        batches = [syn.make_batch(64) if (i == 0 and mix == "full") else syn.make_batch(64, None if mix == "full" else "ragged", seed=4321 + i)
                   for i in range(NREQ)]
        reqs = []
        for b in batches:
            T = max(b.lengths)
            reqs.append(dict(text_emb=_cuda(b.text_emb, dev), init_latents=_cuda(b.init_latents, dev), lengths=b.lengths,
                             latents_out=torch.empty(64, 1, 256, device=dev), joints_out=torch.full((64, T, 22, 3), float("nan"), device=dev)))
        for _ in range(2):                              # the second call replays the captured graph
            big.sample_many(reqs)
        torch.cuda.synchronize()
        assert big.launch_counts()[0] <= 4              # the loop really ran as the persistent launch
        worst = 0.0
        for b, q in zip(batches, reqs):
            T = max(b.lengths)
            joints = torch.empty(64, T, 22, 3, device=dev)
            exact.sample(q["text_emb"], q["init_latents"], b.lengths, None, None, joints)
            torch.cuda.synchronize()
            d = (q["joints_out"] - joints).abs()
            for i, n in enumerate(b.lengths):
                assert torch.isfinite(q["joints_out"][i, :n]).all()
                worst = max(worst, float(d[i, :n].max()))
        report[mix + "_vs_exact_fp32_engine"] = worst
        assert worst < 8e-4, report
        # direct checks against the reference fixture (request 0, every frame) / the CPU oracle: EVERY request of the driver's 1 280-motion call (round 5: no
        # HIP-vs-HIP majority any more), eight requests per length mix of the 2 048-motion call
        for k in (range(NREQ) if NREQ == 20 else ((0, 2, 5, 7, 11, 13, 17, 19) if mix == "full" else (1, 3, 6, 9, 12, 15, 16, 18))):
            b, q = batches[k], reqs[k]
            if mix == "full" and k == 0:
                e = float(np.abs(q["joints_out"].cpu().numpy() - g["joints"]).max())
                assert np.abs(q["latents_out"].cpu().numpy() - g["latents"]).max() < 5e-3
            else:
                jr = ops.to_numpy(O.sample(ops, bd, bv, ops.asarray(b.text_emb), ops.asarray(b.init_latents), b.lengths,
                                           ops.asarray(mean), ops.asarray(std)))
                got = q["joints_out"].cpu().numpy()
                e = max(float(np.abs(got[i, :n] - jr[i, :n]).max()) for i, n in enumerate(b.lengths))
            report[f"{mix}_request{k}_vs_reference"] = e
            assert e < 1e-3, report
    ns = big.numeric_status()
    assert ns["nonfinite_values"] == 0, ns                 # the run-time counter over every latents / joints value of those calls
    report["probe"] = {k: ns[k] for k in ("probe_err_loop", "probe_err_decode")}
    print("headline shape parity:", NREQ, report)
    big.close()
    exact.close()


def test_demo_cli_writes_the_reference_files_on_gpu(dev, tmp_path):
    """python -m mld_hip.demo (the text-to-motion branch of the reference's demo.py:40-50,129-194) end to end on the MI355X: example file
    -> config -> model (offline: synthetic text encoder and weights, announced) -> MLD.forward -> Example_<len>_batch0_<i>.npy/.txt."""
    from mld_hip import demo
    from mld_hip import engine as E
    ex = tmp_path / "example.txt"
    ex.write_text("50 a man kicks with something or someone with his left leg.\n100 A person is skipping rope.\n100 a man bends down and picks something up.\n")
    out = tmp_path / "results"
    E.drop_engines()
    E.configure("text", max_batch=64, max_frames=196, max_in_flight=1)       # the engine defaults of a fresh process (earlier tests shrink them)
    try:
        paths = demo.main(["--example", str(ex), "--out_dir", str(out)])
    finally:
        E.drop_engines()
    assert [os.path.basename(p) for p in paths] == ["Example_50_batch0_0.npy", "Example_100_batch0_1.npy", "Example_100_batch0_2.npy"]
    for p, n in zip(paths, (50, 100, 100)):
        j = np.load(p)
        assert j.shape == (n, 22, 3) and j.dtype == np.float32 and np.isfinite(j).all()
        assert os.path.exists(p.replace(".npy", ".txt"))
    assert open(paths[1].replace(".npy", ".txt")).read() == "A person is skipping rope."


# ---- the range contract of the split-f16 mode (include/mldhip.h "Range contract"; VERDICT r3 "what's weak" 1b / advisor r3 #1)
def _scaled_weights(case):
    """The seeded synthetic weights with one class of tensors pushed out of the half format's comfortable range."""
    sdd, sdv = syn.make_denoiser_state_dict(), syn.make_vae_state_dict()
    if case == "ln_gain_up":          # LayerNorm gains x 2^10: post-norm activations |x| ~ 1e3 .. 4e3
        for sd, keys in ((sdd, ("encoder.input_blocks.1.norm1", "encoder.middle_block.norm2", "encoder.output_blocks.2.norm1")),
                         (sdv, ("decoder.input_blocks.1.norm2", "decoder.output_blocks.0.norm3"))):
            for k in keys:
                sd[k + ".weight"] = sd[k + ".weight"] * np.float32(1024.0)
                sd[k + ".bias"] = sd[k + ".bias"] * np.float32(1024.0)
    elif case == "ln_gain_down":      # LayerNorm gains x 2^-10: |x| ~ 1e-3, high halves near / in the half subnormals, low halves gone
        for sd, keys in ((sdd, ("encoder.input_blocks.1.norm1", "encoder.middle_block.norm2", "encoder.output_blocks.2.norm1")),
                         (sdv, ("decoder.input_blocks.1.norm2", "decoder.output_blocks.0.norm3"))):
            for k in keys:
                sd[k + ".weight"] = sd[k + ".weight"] * np.float32(2.0 ** -10)
                sd[k + ".bias"] = sd[k + ".bias"] * np.float32(2.0 ** -10)
    elif case == "ffn_hidden_huge":   # one feed-forward layer with linear1 x 2^14: pre-GELU |h| ~ 1e4 .. 1e5, beyond 65 504 for some
        for sd, k in ((sdd, "encoder.input_blocks.2"), (sdv, "decoder.input_blocks.2")):
            sd[k + ".linear1.weight"] = sd[k + ".linear1.weight"] * np.float32(2.0 ** 14)
            sd[k + ".linear2.weight"] = sd[k + ".linear2.weight"] * np.float32(2.0 ** -14)      # (keeps the block's output O(1))
    elif case == "weights_tiny":      # weight matrices x 2^-12 (|w| ~ 1e-5: half subnormals), one of them NOT followed by a residual
        sdd["encoder.linear_blocks.0.weight"] = sdd["encoder.linear_blocks.0.weight"] * np.float32(2.0 ** -12)
        sdd["encoder.linear_blocks.0.bias"] = sdd["encoder.linear_blocks.0.bias"] * np.float32(2.0 ** -12)
        sdv["decoder.input_blocks.1.self_attn.in_proj_weight"] = sdv["decoder.input_blocks.1.self_attn.in_proj_weight"] * np.float32(2.0 ** -12)
        sdv["decoder.linear_blocks.1.weight"] = sdv["decoder.linear_blocks.1.weight"] * np.float32(2.0 ** -12)
        sdv["decoder.linear_blocks.1.bias"] = sdv["decoder.linear_blocks.1.bias"] * np.float32(2.0 ** -12)
    else:
        assert case == "plain"
    return sdd, sdv


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["plain", "ln_gain_up", "ln_gain_down", "ffn_hidden_huge", "weights_tiny"])
def test_split_f16_range_contract_out_of_comfort_zone(dev, case):
    """MLDHIP_PREC_F16X3 outside the range the synthetic weights live in (|activation| < ~100, |w| ~ 0.05): LayerNorm gains x 2^+-10,
    a feed-forward layer whose hidden activation passes +-65 504, weight matrices at 1e-5.  The contract (mldhip.h): finalize's
    probe either keeps a stage on the split kernels (probe error <= MLDHIP_PROBE_TOL) or moves it to the exact-fp32 kernels and says
    so; either way the joints stay within 1e-3 of the reference arithmetic (cross_attention.py:259-272, mld.py:290-360 restated by
    the oracle on the SAME modified weights) and nothing non-finite is produced.  Runs the persistent loop (loop_kernel 3, 16
    motions) AND the latency kernels, decoder at 16 x 64 frames."""
    sdd, sdv = _scaled_weights(case)
    mean, std = syn.make_mean_std()
    b = syn.make_batch(16, [64, 57, 64, 33, 64, 64, 12, 64, 40, 64, 64, 64, 25, 64, 64, 1], seed=77)
    ops = O.TorchOps("float32")
    jr = ops.to_numpy(O.sample(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), ops.asarray(b.text_emb), ops.asarray(b.init_latents), b.lengths,
                               ops.asarray(mean), ops.asarray(std)))
    status = {}
    errs = {}
    for name, prec, probe in (("f32", 0, 1), ("x3", 1, 1), ("x3_unguarded", 1, 0)):
        e = _lib.Engine(device=0, max_batch=16, max_frames=64, precision=prec)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae.")
        e.load_tensor("mean", mean); e.load_tensor("std", std)
        e.set_option("range_probe", probe)
        e.finalize()
        for lk in (3, 1):
            e.set_option("loop_kernel", lk)
            joints = torch.full((16, 64, 22, 3), float("nan"), device=dev)
            e.sample(_cuda(b.text_emb, dev), _cuda(b.init_latents, dev), b.lengths, None, None, joints)
            torch.cuda.synchronize()
            got = joints.cpu().numpy()
            errs[(name, lk)] = max(float(np.nan_to_num(np.abs(got[i, :n] - jr[i, :n]), nan=np.inf).max()) for i, n in enumerate(b.lengths))
        status[name] = e.numeric_status()
        e.close()
    print("range contract", case, {k: float("%.3g" % v) for k, v in errs.items()}, {k: v for k, v in status.items() if k != "f32"})
    assert errs[("f32", 3)] < 1e-3 and errs[("f32", 1)] < 1e-3, (case, errs)          # the case itself is well conditioned
    s = status["x3"]
    assert s["probed"] == 1
    for stage in ("loop", "decode"):
        ok, err = s[stage + "_split_ok"], s["probe_err_" + stage]
        assert ok == (1 if err <= _lib.PROBE_TOL else 0), (case, s)                     # the decision is the documented rule
    assert errs[("x3", 3)] < 1e-3 and errs[("x3", 1)] < 1e-3, (case, errs, s)           # the guarded mode keeps the joint contract
    assert s["nonfinite_values"] == 0, (case, s)
    if case == "plain":
        assert s["loop_split_ok"] == 1 and s["decode_split_ok"] == 1, s                 # ... and costs nothing on in-range weights
    if case == "ffn_hidden_huge":
        # the case really is out of range: the probe moved the loop AND the decoder to the exact-fp32 kernels, and without the guard the
        # split kernels are wrong (half saturates / overflows at 65 504) -- by 1e-2 on the joints, or non-finite and then counted
        assert s["loop_split_ok"] == 0 and s["decode_split_ok"] == 0, s
        u = status["x3_unguarded"]
        assert u["probed"] == 0 and (u["nonfinite_values"] > 0 or max(errs[("x3_unguarded", 3)], errs[("x3_unguarded", 1)]) > 1e-3), (errs, u)
