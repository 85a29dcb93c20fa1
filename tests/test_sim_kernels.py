"""CPU checks of the HIP kernels through the TEST-ONLY functional simulator (tests/hipemu).

The same kernel sources that hipcc compiles for gfx950 are compiled for the host with -DMLDHIP_SIM
and run wave-by-wave (64-lane fibers, exact v_mfma_f32_16x16x4_f32 model).  This proves the index
math / masking / fusion logic against the oracle without a GPU; the GPU suite (-m gpu) is the parity
test proper.  Nothing under motion-latent-diffusion_amd/ loads the simulator.
"""
import numpy as np
import pytest

import simlib
from mld_hip import _lib
from mld_hip import synthetic as syn
from oracle import mld_oracle as O


@pytest.fixture(scope="module")
def eng():
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2)
    yield e
    e.close()


@pytest.fixture(scope="module")
def ow():
    ops = O.NumpyOps(np.float32)
    sdd, sdv = simlib.text_weights()
    return ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv)


def test_schedule_tables(eng):
    sch = O.DDIMSchedule()
    np.testing.assert_array_equal(eng.timesteps(), sch.set_timesteps(2))
    np.testing.assert_allclose(eng.alphas_cumprod(), sch.alphas_cumprod, rtol=2e-6)


def test_ignored_and_required_keys():
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=16)
    n_required = len(e.missing_keys())
    # denoiser sans mem_pos, vae decoder, mean/std, vae encoder (token, pe, skel_embedding, 9 layers, skips, norm)
    assert n_required == (126 - 1) + (1 + 9 * 18 + 8 + 2 + 2) + 2 + (1 + 1 + 2 + 9 * 12 + 8 + 2)
    ignored = simlib.load_synthetic_weights(e, finalize=False)
    assert e.missing_keys() == []
    assert ignored == ["denoiser.mem_pos.pe"]                  # the only checkpoint tensor no kernel reads
    e.finalize()
    e.close()


def test_denoiser_forward_sim(eng, ow):
    ops, bd, _ = ow
    b = syn.make_batch(2, [20, 13])
    x = np.concatenate([b.init_latents] * 2)
    out = np.zeros((4, 1, 256), np.float32)
    eng.denoiser_forward(x, 981, b.text_emb, 4, out)
    ref = O.denoiser_forward(ops, bd, x, 981, b.text_emb)
    assert np.abs(out - ref).max() < 5e-5


def test_denoiser_forward_odd_row_count_sim(eng, ow):
    """R=3 -> 9 token rows: exercises row clamping in the 16-row MFMA tiles and the idle attention lanes."""
    ops, bd, _ = ow
    b = syn.make_batch(3, [8, 8, 8], seed=3)
    x = b.init_latents
    out = np.zeros((3, 1, 256), np.float32)
    eng.denoiser_forward(x, 37, b.text_emb[3:], 3, out)
    ref = O.denoiser_forward(ops, bd, x, 37, b.text_emb[3:])
    assert np.abs(out - ref).max() < 5e-5


def test_vae_decode_ragged_sim(eng, ow):
    ops, _, bv = ow
    lengths = [20, 13, 1]
    z = syn._rng(3, "z").standard_normal((3, 1, 256)).astype(np.float32)
    feats = np.full((3, 20, 263), np.nan, np.float32)
    eng.vae_decode(z, lengths, feats)
    ref = O.vae_decode(ops, bv, z, lengths)
    assert np.isfinite(feats).all()
    assert np.abs(feats - ref).max() < 5e-5
    assert (feats[1, 13:] == 0).all() and (feats[2, 1:] == 0).all()


def test_vae_encode_sim(eng, ow):
    """Scope row 8f.1: MldVae.encode on the decoder's kernels (T+2 tokens, key-padding mask, padded K = 263)."""
    ops, _, bv = ow
    lengths, T = [20, 13, 1], 22                      # padded length > max(lengths) is legal
    fe = syn._rng(9, "f").standard_normal((3, T, 263)).astype(np.float32)
    for i, n in enumerate(lengths):
        fe[i, n:] = 0
    eps = syn._rng(10, "e").standard_normal((3, 1, 256)).astype(np.float32)
    lat, mu, lv = (np.zeros((3, 1, 256), np.float32) for _ in range(3))
    eng.vae_encode(fe, lengths, T, eps, lat, mu, lv)
    lr, mr, lvr = O.vae_encode(ops, bv, fe, lengths, eps)
    assert np.abs(mu - mr).max() < 5e-5 and np.abs(lv - lvr).max() < 5e-5 and np.abs(lat - lr).max() < 1e-4
    with pytest.raises(_lib.MldHipError):
        eng.vae_encode(fe, [30, 13, 1], T, eps, lat, mu, lv)     # a length beyond the padded T


def test_feats2joints_sim(eng, ow):
    ops = ow[0]
    f = syn._rng(8, "f").standard_normal((2, 37, 263)).astype(np.float32)
    mean, std = syn.make_mean_std()
    j = np.zeros((2, 37, 22, 3), np.float32)
    eng.feats2joints(f, 2, 37, j)
    assert np.abs(j - O.feats2joints(ops, f, mean, std)).max() < 2e-5


def test_ddim_step_sim(eng):
    sch = O.DDIMSchedule()
    sch.set_timesteps(2)
    e = syn._rng(4, "e").standard_normal((2, 256)).astype(np.float32)
    x = syn._rng(5, "x").standard_normal((2, 256)).astype(np.float32)
    o = np.zeros_like(x)
    eng.ddim_step(e, 501, x, o, x.size)
    assert np.abs(o - sch.step(e, 501, x)).max() < 1e-6


def test_full_sample_sim(eng, ow):
    ops, bd, bv = ow
    b = syn.make_batch(2, [20, 13])
    mean, std = syn.make_mean_std()
    lat = np.zeros((2, 1, 256), np.float32)
    feats = np.zeros((2, 20, 263), np.float32)
    joints = np.zeros((2, 20, 22, 3), np.float32)
    eng.sample(b.text_emb, b.init_latents, b.lengths, lat, feats, joints)
    jr, fr, lr = O.sample(ops, bd, bv, b.text_emb, b.init_latents, b.lengths, mean, std, steps=2, return_intermediates=True)
    assert np.abs(lat - lr).max() < 5e-4
    assert np.abs(feats - fr).max() < 1e-4
    assert np.abs(joints - jr).max() < 1e-4
    den, dec, jn = eng.launch_counts()
    # text projection + per chain (default: one chain): init + steps * (L layers * 4 + (L - 1) / 2 skip + 1 final)
    L, nb = simlib.SIM_LAYERS, (simlib.SIM_LAYERS - 1) // 2
    assert den == 1 + 1 * (1 + 2 * (L * 4 + nb + 1)) and dec == 2 + 1 + L * 5 + nb + 2 and jn == 1


def test_abi_errors_sim(eng):
    b = syn.make_batch(2, [10, 10])
    with pytest.raises(_lib.MldHipError) as ei:
        eng.sample(b.text_emb, b.init_latents, [10, 41], None, None, None)
    assert "max_frames" in str(ei.value)
    with pytest.raises(_lib.MldHipError):
        eng.sample(b.text_emb, b.init_latents, [10, 0], None, None, None)
    with pytest.raises(_lib.MldHipError):
        eng.denoiser_forward(b.init_latents, -1, b.text_emb, 2, b.init_latents)
    with pytest.raises(_lib.MldHipError):
        _lib.Engine(lib=simlib.sim_library(), latent_dim=512)
    with pytest.raises(_lib.MldHipError):
        _lib.Engine(lib=simlib.sim_library(), num_layers=8)


# ------------------------------------------------------------------ action-conditioned variant (BASELINE config 5)
@pytest.fixture(scope="module")
def aeng():
    e = simlib.sim_action_engine(max_batch=4, max_frames=24, num_inference_steps=2)
    yield e
    e.close()


@pytest.fixture(scope="module")
def aow():
    ops = O.NumpyOps(np.float32)
    sdd, sdv = simlib.action_weights()
    return ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv)


def test_action_required_keys_sim():
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=16, **simlib.ACTION_CFG)
    # denoiser: time MLP 4 + table 1 + pe 1 + 15 layers * 12 + 7 skip linears * 2 + norm 2; ActorVae decoder: pe + 6 * 18 + final 2;
    # ActorVae encoder: 2 tokens + pe + skel_embedding 2 + 6 * 12; mean/std
    assert len(e.missing_keys()) == (4 + 1 + 1 + 15 * 12 + 14 + 2) + (1 + 6 * 18 + 2) + (2 + 1 + 2 + 6 * 12) + 2
    ignored = simlib.load_action_weights(e, finalize=False)
    assert e.missing_keys() == ["mean", "std"]                 # optional group (no joints on this layout)
    assert ignored == ["denoiser.mem_pos.pe"]
    e.finalize()
    with pytest.raises(_lib.MldHipError):                      # text entry point on an action engine
        e.sample(np.zeros((4, 1, 768), np.float32), np.zeros((2, 1, 256), np.float32), [8, 8], None, None, None)
    with pytest.raises(_lib.MldHipError):                      # label out of range
        e.sample_action([0, 12], np.zeros((2, 1, 256), np.float32), [8, 8], None, None)
    e.close()


def test_action_denoiser_forward_sim(aeng, aow):
    ops, bd, _ = aow
    acts, lat0, _ = syn.make_action_batch(2, 16)
    x = np.concatenate([lat0, lat0])
    cond = np.concatenate([np.zeros_like(acts), acts])
    out = np.zeros((4, 1, 256), np.float32)
    aeng.denoiser_forward_action(x, 981, cond, out)
    ref = O.denoiser_forward_action(ops, bd, x, 981, cond)
    assert np.abs(out - ref).max() < 5e-5
    # the unconditional half ignores its labels (EmbedAction zeroes it, mld_denoiser.py:253-257)
    out2 = np.zeros_like(out)
    aeng.denoiser_forward_action(x, 981, np.concatenate([acts[::-1], acts]), out2)
    np.testing.assert_array_equal(out, out2)


def test_actor_decode_ragged_sim(aeng, aow):
    ops, _, bv = aow
    z = syn._rng(5, "az").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [24, 9, 17]
    feats = np.full((3, 24, 150), 7.0, np.float32)
    aeng.vae_decode(z, lens, feats)
    ref = O.actor_decode(ops, bv, z, lens)
    assert np.abs(feats - ref).max() < 5e-5
    assert np.all(feats[1, 9:] == 0) and np.all(feats[2, 17:] == 0)


def test_action_full_sample_sim(aeng, aow):
    ops, bd, bv = aow
    acts, lat0, _ = syn.make_action_batch(3, 16)
    lens = [16, 16, 11]
    lat = np.zeros((3, 1, 256), np.float32)
    feats = np.zeros((3, 16, 150), np.float32)
    aeng.sample_action(acts, lat0, lens, lat, feats)
    fr, lr = O.sample_action(ops, bd, bv, acts, lat0, lens, steps=2, return_intermediates=True)
    assert np.abs(lat - lr).max() < 5e-4
    assert np.abs(feats - fr).max() < 1e-4
    den, dec, _ = aeng.launch_counts()
    # label gather + init + steps * (15 layers * 4 + 7 skip + 1 final); decode: 2 cross-attn + queries + 6 * 5 + final
    La, Lv = simlib.SIM_ACTION_LAYERS, simlib.SIM_ACTOR_VAE_LAYERS
    assert den == 1 + 1 + 2 * (La * 4 + (La - 1) // 2 + 1) and dec == 2 + 1 + Lv * 5 + 1


# ------------------------------------------------------------------ diffusion-only variant (BASELINE config 4)
@pytest.fixture(scope="module")
def neng():
    e = simlib.sim_novae_engine(num_layers=2, max_batch=2, max_frames=40, num_inference_steps=4)
    yield e
    e.close()


@pytest.fixture(scope="module")
def now():
    ops = O.NumpyOps(np.float32)
    return ops, O.to_backend(ops, syn.make_novae_denoiser_state_dict(dims=syn.ModelDims(latent_dim=512, num_layers=2)))


def test_novae_schedule_and_keys_sim(neng):
    sch = O.DDPMSchedule()
    np.testing.assert_array_equal(neng.timesteps(), sch.set_timesteps(4))          # [750, 500, 250, 0]
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=16, **simlib.NOVAE_CFG)
    # pose_embd/pose_proj 4, time MLP 4, emb_proj 2, two PEs, 9 decoder layers x 18, final norm 2, mean/std
    assert len(e.missing_keys()) == 4 + 4 + 2 + 2 + 9 * 18 + 2 + 2
    e.close()
    with pytest.raises(_lib.MldHipError):                      # unsupported combination fails at create
        _lib.Engine(lib=simlib.sim_library(), latent_dim=512, vae_arch=_lib.VAE_NONE)
    with pytest.raises(_lib.MldHipError):                      # text-latent entry point on a diffusion-only engine
        neng.sample(np.zeros((4, 1, 768), np.float32), np.zeros((2, 1, 256), np.float32), [8, 8], None, None, None)


def test_novae_denoiser_forward_sim(neng, now):
    ops, bd = now
    g = syn._rng(11, "nv")
    R, T = 3, 21                                               # ragged tiles: 63 rows, T not a multiple of 16
    x = g.standard_normal((R, T, 263)).astype(np.float32)
    te = g.standard_normal((R, 1, 768)).astype(np.float32)
    lens = [21, 13, 0]
    out = np.full((R, T, 263), 7.0, np.float32)
    neng.denoiser_forward_novae(x, 999, te, lens, T, out)
    ref = O.denoiser_forward_novae(ops, bd, x, 999, te, lens)
    assert np.abs(out - ref).max() < 5e-5
    assert np.all(out[1, 13:] == 0) and np.all(out[2] == 0)


def test_ddpm_step_and_philox_sim(neng):
    sch = O.DDPMSchedule()
    sch.set_timesteps(4)
    g = syn._rng(12, "dd")
    x, eps, nz = (g.standard_normal(1001).astype(np.float32) for _ in range(3))
    for t in (750, 250, 0):
        o = np.zeros_like(x)
        neng.ddpm_step(eps, t, x, nz, o, x.size)
        assert np.abs(o - sch.step(eps, t, x, nz)).max() < 2e-6
    z = np.zeros(1001, np.float32)
    neng.philox_normal(z, z.size, 0x1234567890ABCDEF, 7)
    zr = O.philox_normal(z.size, 0x1234567890ABCDEF, 7)
    assert np.abs(z - zr).max() < 2e-5                          # same Philox bits; libm log/cos/sin differ in the last ulps
    o1, o2 = np.zeros_like(x), np.zeros_like(x)
    neng.ddpm_step(eps, 500, x, None, o1, x.size, seed=0x1234567890ABCDEF, step_index=7)   # in-kernel noise == the exposed stream
    neng.ddpm_step(eps, 500, x, z, o2, x.size)
    assert np.abs(o1 - o2).max() < 1e-6


def test_novae_full_sample_sim(neng, now):
    ops, bd = now
    B, T = 2, 20
    b = syn.make_batch(B, [20, 11])
    g = syn._rng(13, "nvs")
    lat0 = g.standard_normal((B, T, 263)).astype(np.float32)
    noise = g.standard_normal((4, B, T, 263)).astype(np.float32)
    mean, std = syn.make_mean_std()
    feats = np.zeros((B, T, 263), np.float32)
    joints = np.zeros((B, T, 22, 3), np.float32)
    neng.sample_novae(b.text_emb, lat0, b.lengths, noise, 0, feats, joints)
    jr, fr = O.sample_novae(ops, bd, b.text_emb, lat0, b.lengths, noise, mean, std, steps=4)
    assert np.abs(feats - fr).max() < 2e-4
    assert np.abs(joints - jr).max() < 2e-4
    den, dec, jn = neng.launch_counts()
    # text memory 2 + steps * (pad + embed 2 + 2 layers * 11 + final norm/proj 2 + step)
    assert den == 3 + 4 * (1 + 2 + 2 * 7 + 2 + 1) and dec == 0 and jn == 1      # (7 launches per layer since "cross_fold": 11 before; + the text tokens' fold in the prologue)


def test_novae_denoiser_forward_staged_gemms_sim(now):
    """Same check with the LDS-staged GEMM pipeline forced (production shape: K = 384 / 512 / 1024 chunk pipelines)."""
    ops, bd = now
    e = simlib.sim_novae_engine(num_layers=2, max_batch=2, max_frames=40, num_inference_steps=4)
    e.set_option("gemm_small_m", 0)
    g = syn._rng(11, "nv")
    R, T = 3, 21
    x = g.standard_normal((R, T, 263)).astype(np.float32)
    te = g.standard_normal((R, 1, 768)).astype(np.float32)
    lens = [21, 13, 5]
    out = np.zeros((R, T, 263), np.float32)
    e.denoiser_forward_novae(x, 999, te, lens, T, out)
    ref = O.denoiser_forward_novae(ops, bd, x, 999, te, lens)
    assert np.abs(out - ref).max() < 5e-5
    e.close()


def test_contexts_rotate_and_stay_exact_sim(ow):
    """max_in_flight = 2: consecutive calls alternate between two workspaces (shared weights); results must not depend on it."""
    ops, bd, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2, max_in_flight=2)
    mean, std = syn.make_mean_std()
    for seed, lens in ((1, [20, 13]), (2, [9, 24, 17]), (3, [5])):
        b = syn.make_batch(len(lens), lens, seed=seed)
        joints = np.zeros((len(lens), max(lens), 22, 3), np.float32)
        e.sample(b.text_emb, b.init_latents, b.lengths, None, None, joints)
        jr = O.sample(ops, bd, bv, b.text_emb, b.init_latents, b.lengths, mean, std, steps=2)
        assert np.abs(joints - jr).max() < 1e-4
    with pytest.raises(_lib.MldHipError):
        _lib.Engine(lib=simlib.sim_library(), max_in_flight=9)
    e.close()


def test_staged_gemm_tiles_are_exact_sim(ow):
    """The staged fp32 GEMM tiles on 8 waves (64x128 plain, 64x256 with the LayerNorm epilogue), forced at simulator-sized M."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2)
    e.set_option("gemm_small_m", 0)                     # force the staged kernels at simulator-sized M
    z = syn._rng(7, "g8").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [40, 23, 7]
    feats = np.zeros((3, 40, 263), np.float32)
    e.vae_decode(z, lens, feats)
    assert np.abs(feats - O.vae_decode(ops, bv, z, lens)).max() < 5e-5
    e.close()


def test_actor_encode_sim(aeng, aow):
    ops, _, bv = aow
    g = syn._rng(9, "actor_feats_sim")
    fe = g.standard_normal((3, 20, 150)).astype(np.float32)
    lens = [20, 11, 3]
    for i, n in enumerate(lens):
        fe[i, n:] = 0
    eps = g.standard_normal((3, 256)).astype(np.float32)
    lat, mu, lv = (np.zeros((3, 256), np.float32) for _ in range(3))
    aeng.vae_encode(fe, lens, 20, eps, lat, mu, lv)
    lr, mr, lvr = O.actor_encode(ops, bv, fe, lens, eps[:, None, :])
    assert np.abs(mu - mr[:, 0]).max() < 5e-5 and np.abs(lv - lvr[:, 0]).max() < 5e-5
    assert np.abs(lat - lr[:, 0]).max() < 1e-4


def test_split_bf16_decoder_gemms_sim(ow):
    """precision = BF16X3_DECODE on the simulator's v_mfma_f32_16x16x32_bf16 model: x = hi + lo in bf16, three MFMAs per
    K chunk (hi*hi + hi*lo + lo*hi), fp32 accumulate -- decoder features stay within ~1e-4 of the fp32 oracle."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)                     # force the staged (split-capable) kernels at simulator-sized M
    z = syn._rng(7, "g8").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [40, 23, 7]
    feats = np.zeros((3, 40, 263), np.float32)
    e.vae_decode(z, lens, feats)
    ref = O.vae_decode(ops, bv, z, lens)
    err = np.abs(feats - ref).max()
    assert 1e-7 < err < 2e-4                            # not bit-equal to fp32 (the split really ran), yet well inside tolerance
    e.close()


# ------------------------------------------------------------------------------------------------------------------
# throughput kernel family (kernels/strip.hpp + the 32x64 staged GEMM): same data flow, one raw slab per GEMM
def test_strip_family_sample_is_exact_sim(ow):
    """loop_kernel = 2 forces the throughput kernels at simulator-sized M: ragged batch (M = 18 rows: one partial strip),
    2 steps, against the oracle; launch counts are those of the latency family."""
    ops, bd, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2)
    e.set_option("loop_kernel", 2)
    b = syn.make_batch(3, [20, 13, 7])
    mean, std = syn.make_mean_std()
    lat = np.zeros((3, 1, 256), np.float32)
    joints = np.zeros((3, 20, 22, 3), np.float32)
    e.sample(b.text_emb, b.init_latents, b.lengths, lat, None, joints)
    jr, _, lr = O.sample(ops, bd, bv, b.text_emb, b.init_latents, b.lengths, mean, std, steps=2, return_intermediates=True)
    assert np.abs(lat - lr).max() < 5e-4 and np.abs(joints - jr).max() < 1e-4
    assert e.launch_counts()[0] == 1 + 1 + 2 * (simlib.SIM_LAYERS * 4 + (simlib.SIM_LAYERS - 1) // 2 + 1)
    with pytest.raises(_lib.MldHipError):
        e.set_option("loop_kernel", 4)
    with pytest.raises(_lib.MldHipError):
        e.set_option("no_such_option", 1)
    e.close()


def test_strip_family_denoiser_forward_matches_latency_family_sim(ow):
    """One MldDenoiser.forward at R = 22 (M = 66 rows: two full strips + a partial one) on both kernel families and vs the
    oracle; auto mode switches families at strip_min_rows."""
    ops, bd, _ = ow
    e = simlib.sim_engine(max_batch=11, max_frames=8, num_inference_steps=2)
    g = syn._rng(21, "strip")
    R = 22
    x = g.standard_normal((R, 1, 256)).astype(np.float32)
    te = g.standard_normal((R, 1, 768)).astype(np.float32)
    ref = np.asarray(O.denoiser_forward(ops, bd, x, 741, te))
    outs = {}
    for name, opts in (("latency", {"loop_kernel": 1}), ("throughput", {"loop_kernel": 2}), ("auto", {"loop_kernel": 0, "strip_min_rows": 60})):
        for k, v in opts.items():
            e.set_option(k, v)
        out = np.zeros((R, 1, 256), np.float32)
        e.denoiser_forward(x, 741, te, R, out)
        assert np.abs(out - ref).max() < 5e-5, name
        outs[name] = out
    assert np.array_equal(outs["auto"], outs["throughput"])            # 66 rows >= 60: auto picked the throughput kernels
    assert np.abs(outs["latency"] - outs["throughput"]).max() < 2e-5   # different summation order only
    e.close()


def test_strip_family_action_variant_sim(aow):
    """15-layer action denoiser (7 skip linears: the two-segment K = 512 strip kernel) on the throughput family."""
    ops, bd, bv = aow
    e = simlib.sim_action_engine(max_batch=4, max_frames=20, num_inference_steps=2)
    e.set_option("loop_kernel", 2)
    acts, lat0, lens = syn.make_action_batch(3, nframes=20, seed=5)
    lat = np.zeros((3, 1, 256), np.float32)
    feats = np.zeros((3, 20, 150), np.float32)
    e.sample_action(acts, lat0, lens, lat, feats)
    fr, lr = O.sample_action(ops, bd, bv, acts, lat0, lens, steps=2, return_intermediates=True)
    assert np.abs(lat - np.asarray(lr)).max() < 5e-4 and np.abs(feats - np.asarray(fr)).max() < 2e-4
    e.close()


def test_sample_many_equals_per_request_calls_sim(ow):
    """mldhip_sample_many: three requests (ragged, different Tmax, one without joints) coalesced into one chain give every
    request what its own mldhip_sample call gives (same kernel family forced, so bit-identical here)."""
    e = simlib.sim_engine(max_batch=6, max_frames=24, num_inference_steps=2)
    e.set_option("loop_kernel", 1)
    reqs, solo = [], []
    for seed, lens in ((1, [20, 13]), (2, [9]), (3, [24, 5, 17])):
        b = syn.make_batch(len(lens), lens, seed=seed)
        B, T = len(lens), max(lens)
        q = dict(text_emb=b.text_emb, init_latents=b.init_latents, lengths=lens, latents_out=np.zeros((B, 1, 256), np.float32),
                 feats_out=np.zeros((B, T, 263), np.float32), joints_out=None if seed == 2 else np.zeros((B, T, 22, 3), np.float32))
        reqs.append(q)
        lat, feats, joints = np.zeros((B, 1, 256), np.float32), np.zeros((B, T, 263), np.float32), np.zeros((B, T, 22, 3), np.float32)
        e.sample(b.text_emb, b.init_latents, lens, lat, feats, joints)
        solo.append((lat, feats, joints))
    e.sample_many(reqs)
    for q, (lat, feats, joints) in zip(reqs, solo):
        assert np.array_equal(q["latents_out"], lat) and np.array_equal(q["feats_out"], feats)
        if q["joints_out"] is not None:
            assert np.array_equal(q["joints_out"], joints)
    with pytest.raises(_lib.MldHipError):
        e.sample_many(reqs + reqs)                                   # 12 motions > max_batch 6
    e.close()


def test_sample_many_action_sim(aow):
    ops, bd, bv = aow
    e = simlib.sim_action_engine(max_batch=4, max_frames=20, num_inference_steps=2)
    reqs = []
    for seed, n in ((5, 2), (6, 1)):
        acts, lat0, lens = syn.make_action_batch(n, nframes=20, seed=seed)
        reqs.append(dict(actions=acts, init_latents=lat0, lengths=lens, feats_out=np.zeros((n, max(lens), 150), np.float32)))
    e.sample_many(reqs)
    for q in reqs:
        fr = O.sample_action(ops, bd, bv, q["actions"], q["init_latents"], q["lengths"], steps=2)
        assert np.abs(q["feats_out"] - np.asarray(fr)).max() < 2e-4
    e.close()


# ------------------------------------------------------------------------------------------------------------------
# reduced-precision operand formats (mldhip.h MLDHIP_PREC_BF16; the fp8 mode was retired in round 6) on the simulator's model of
# v_mfma_f32_16x16x32_bf16: the point here is that the kernels' fragment indexing and scaling are right -- the
# result must be CLOSE to the fp32 oracle (error of the format, not garbage) and NOT equal to it (the mode really ran).
# (precision 1 = split-f16: the latency kernels run three f16 MFMAs per K chunk on 22-bit operands -- "tile_x3", on by default)
@pytest.mark.parametrize("prec,fam,lo,hi", [(1, 1, 1e-8, 1e-4), (2, 1, 1e-6, 5e-2), (2, 2, 1e-6, 5e-2)])
def test_reduced_precision_loop_kernels_sim(ow, prec, fam, lo, hi):
    ops, bd, _ = ow
    e = simlib.sim_engine(max_batch=11, max_frames=8, num_inference_steps=2, precision=prec)
    e.set_option("loop_kernel", fam)
    g = syn._rng(21, "strip")
    R = 22
    x = g.standard_normal((R, 1, 256)).astype(np.float32)
    te = g.standard_normal((R, 1, 768)).astype(np.float32)
    ref = np.asarray(O.denoiser_forward(ops, bd, x, 741, te))
    out = np.zeros((R, 1, 256), np.float32)
    e.denoiser_forward(x, 741, te, R, out)
    err = np.abs(out - ref).max()
    print("prec", prec, "family", fam, "denoiser max-abs err", err, "ref max", np.abs(ref).max())
    assert lo < err < hi
    e.close()


def test_bf16_mode_decoder_gemms_sim(ow):
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2, precision=2)
    e.set_option("gemm_small_m", 0)
    z = syn._rng(7, "g8").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [40, 23, 7]
    feats = np.zeros((3, 40, 263), np.float32)
    e.vae_decode(z, lens, feats)
    err = np.abs(feats - O.vae_decode(ops, bv, z, lens)).max()
    print("bf16 decoder feats err", err)
    assert 1e-5 < err < 5e-2
    e.close()


def test_split_bf16_attention_odd_key_tiles_sim(ow):
    """attn_decode_x3_kernel with an ODD number of key tiles (T = 100 -> 7 tiles: the last 32-key block of P.V is half empty)
    and ragged lengths, inside the split-bf16 decoder: features within 2e-4 of the fp32 oracle."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=2, max_frames=100, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    z = syn._rng(8, "x3attn").standard_normal((2, 1, 256)).astype(np.float32)
    lens = [100, 37]
    feats = np.zeros((2, 100, 263), np.float32)
    e.vae_decode(z, lens, feats)
    err = np.abs(feats - O.vae_decode(ops, bv, z, lens)).max()
    print("x3 decoder (7 key tiles) feats err", err)
    assert 1e-7 < err < 2e-4
    assert np.all(feats[1, 37:] == 0)
    e.close()


def test_novae_split_bf16_gemms_and_attention_sim(now):
    """Diffusion-only denoiser with precision = BF16X3_DECODE: every staged GEMM split-bf16 and attn_seq_x3_kernel (head dim 128,
    K then V^T planes through one LDS buffer; T = 37 -> 3 key tiles, an odd count) -- within 3e-4 of the fp32 oracle, not equal."""
    ops, bd = now
    e = simlib.sim_novae_engine(num_layers=2, max_batch=2, max_frames=40, num_inference_steps=4, precision=1)
    e.set_option("gemm_small_m", 0)
    g = syn._rng(12, "nvx3")
    R, T = 2, 37
    x = g.standard_normal((R, T, 263)).astype(np.float32)
    te = g.standard_normal((R, 1, 768)).astype(np.float32)
    lens = [37, 20]
    out = np.zeros((R, T, 263), np.float32)
    e.denoiser_forward_novae(x, 999, te, lens, T, out)
    err = np.abs(out - O.denoiser_forward_novae(ops, bd, x, 999, te, lens)).max()
    print("novae x3 denoiser err", err)
    assert 1e-7 < err < 3e-4
    # "gemm_pipe" = 2: the K = 512 / 1024 GEMMs (in-projection N = 1536, the three d x d projections, linear1 + GELU N = 1024, linear2 K = 1024)
    # on the software-pipelined 128 x 256 tile (kernels/gemm_pipe.hpp: fragment reloads under the matrix instructions, transposed products,
    # XCD-aware 1-D grid whose surplus workgroups exit) at M = 74 rows -- one ragged row tile.  Same products in the same order as the
    # 64 x 128 staged tile the run above used (the default, 1, takes the big tile from 2 048 rows): identical to the bit.
    e.set_option("gemm_pipe", 2)
    out2 = np.zeros((R, T, 263), np.float32)
    e.denoiser_forward_novae(x, 999, te, lens, T, out2)
    assert np.array_equal(out, out2)
    # "flash_attn" = 2: the head-dim-128 form of the key-blocked attention (attention.hpp attn_flash128_x3_kernel: 32-key blocks double buffered,
    # online softmax with the lazy reference point, V through the transpose read, 4 contraction chunks / 8 output dim tiles) instead of
    # attn_seq_x3_kernel; T = 37 -> two key blocks, the second one crossing the sequence end.  Another summation order: close, not equal.
    e.set_option("flash_attn", 2)
    out3 = np.zeros((R, T, 263), np.float32)
    e.denoiser_forward_novae(x, 999, te, lens, T, out3)
    err3 = np.abs(out3 - O.denoiser_forward_novae(ops, bd, x, 999, te, lens)).max()
    d3 = np.abs(out3 - out).max()
    print("novae x3 denoiser err with the key-blocked attention", err3, "difference to the two-phase kernel", d3)
    assert 1e-7 < err3 < 3e-4 and 0 < d3 < 2e-4
    e.close()


def test_staged_split_f16_gemms_vs_oracle_sim(ow, aow):
    """precision = F16X3 with the register-direct kernels off ("ffn_strip" = 0, "strip_gemm" = 0): the LDS-staged GEMM family of gemm.hpp on
    split-f16 MFMAs reading the pre-split weight image finalize builds (elementwise.hpp split_bf16_weights_kernel) -- the path small
    launches and the encoder take.  Decoder with ragged lengths (80 rows: one full 64-row tile, one partial, padded-frame skipping on),
    the MldVae encoder (S = T + 2 rows per sample) and the ActorVae decoder, each against the oracle.  (Round 3 also carried an in-kernel
    weight split and an LDS-staged fused feed-forward kernel as A/B knobs; both retired in round 4.)"""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    e.set_option("ffn_strip", 0)
    e.set_option("strip_gemm", 0)
    z = syn._rng(7, "g8").standard_normal((2, 1, 256)).astype(np.float32)
    feats = np.zeros((2, 40, 263), np.float32)
    e.vae_decode(z, [40, 23], feats)
    err = np.abs(feats - O.vae_decode(ops, bv, z, [40, 23])).max()
    assert 1e-7 < err < 2e-4
    e.close()
    aops, _, abv = aow
    e = simlib.sim_action_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    e.set_option("ffn_strip", 0)
    e.set_option("strip_gemm", 0)
    za = syn._rng(9, "adec").standard_normal((2, 1, 256)).astype(np.float32)
    f = np.zeros((2, 24, 150), np.float32)
    e.vae_decode(za, [24, 11], f)
    assert np.abs(f - O.actor_decode(aops, abv, za, [24, 11])).max() < 2e-4
    e.close()


def test_key_blocked_attention_matches_whole_kv_attention_sim(ow):
    """attn_flash_x3_kernel ("flash_attn" = 2: online softmax over 32-key blocks, both query tiles of a wave per block) against
    attn_decode_x3_kernel (= 0: all of K / V of a (sample, head) in LDS) inside the split-bf16 decoder: ragged lengths, an odd number
    of key tiles (T = 68 -> 5), lengths that are not multiples of 16, more query tiles than waves (130 frames -> 9).  Different
    summation order and an unnormalised P operand: features within 5e-5 of each other and both within 2e-4 of the fp32 oracle."""
    ops, _, bv = ow
    for B, T, lens in ((1, 68, [37]),):       # (130 of 132 frames, 9 query tiles: test_key_blocked_attention_moves_its_reference_point_sim)
        e = simlib.sim_engine(max_batch=B, max_frames=T, num_inference_steps=2, precision=1)
        e.set_option("gemm_small_m", 0)
        z = syn._rng(8, "x3attn").standard_normal((B, 1, 256)).astype(np.float32)
        ref = np.asarray(O.vae_decode(ops, bv, z, lens))
        outs = []
        for fl in (2, 0):
            e.set_option("flash_attn", fl)
            feats = np.zeros((B, T, 263), np.float32)
            e.vae_decode(z, lens, feats)
            assert 1e-7 < np.abs(feats[:, :max(lens)] - ref).max() < 2e-4
            for i, n in enumerate(lens):
                assert np.all(feats[i, n:] == 0)
            outs.append(feats)
        d = np.abs(outs[0] - outs[1]).max()
        assert 0 < d < 5e-5, d
        e.close()


def test_key_blocked_attention_moves_its_reference_point_sim():
    """The online softmax of attn_flash_x3_kernel keeps a LAZY reference (it moves only when a key block's maximum exceeds it by more
    than 8 in the log2 domain).  With the synthetic weights the scores of a query differ by less than that, so only the first block
    ever takes the rescale branch; here the self-attention in-projections are scaled x12 (score spread ~ x144): references move in
    later blocks too, P reaches its 2^8 bound, and the result must still match the whole-K/V kernel and the oracle."""
    sdd, sdv = simlib.text_weights()
    sdv = dict(sdv)
    scaled = 0
    for k in sdv:
        if "self_attn.in_proj_weight" in k and k.startswith("decoder."):
            w = sdv[k].copy()
            w[:512] *= 12.0                       # q and k rows
            sdv[k] = w
            scaled += 1
    assert scaled > 0
    ops = O.NumpyOps(np.float32)
    bv = O.to_backend(ops, sdv)
    B, T, lens = 1, 132, [130]
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=B, max_frames=T, num_inference_steps=2, num_layers=simlib.SIM_LAYERS, precision=1)
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(sdv, "vae.")
    mean, std = syn.make_mean_std()
    e.load_tensor("mean", mean)
    e.load_tensor("std", std)
    e.finalize()
    e.set_option("gemm_small_m", 0)
    z = syn._rng(8, "x3attn").standard_normal((B, 1, 256)).astype(np.float32)
    ref = np.asarray(O.vae_decode(ops, bv, z, lens))
    outs = []
    for fl in (2, 0):
        e.set_option("flash_attn", fl)
        feats = np.zeros((B, T, 263), np.float32)
        e.vae_decode(z, lens, feats)
        assert np.isfinite(feats).all() and np.abs(feats[:, :130] - ref).max() < 5e-4
        outs.append(feats)
    assert 0 < np.abs(outs[0] - outs[1]).max() < 1e-4
    e.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_sample_major_persistent_loop_sim(prec):
    """loop_kernel = 3 (kernels/loop_fused.hpp): the whole reverse loop as ONE launch, a workgroup per 8 motions -- B = 11 (two
    workgroups, the second one with 3 live motions and 5 clamped duplicates), a 3-layer skip stack (one skip linear), 2 steps,
    against the oracle and against the latency family; exact-fp32 MFMAs (precision 0, and precision 1 with fused_x3 = 0) and the
    split-f16 MFMAs of precision 1 (row-swizzled operand images: test_lds_layout.py proves the map).  Launch count: condition rows + the loop."""
    dims = syn.ModelDims(num_layers=3)
    sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=12, max_frames=8, num_inference_steps=2, num_layers=3, precision=prec)
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(sdv, "vae.")
    e.finalize()
    b = syn.make_batch(11, [8, 5, 3, 8, 1, 7, 2, 6, 8, 4, 8], seed=9)      # lengths do not matter to the loop (latents only)
    ops = O.NumpyOps(np.float32)
    ref = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, sdd), b.text_emb, b.init_latents, 7.5, 2, 4))
    e.set_option("loop_kernel", 1)
    lat1 = np.zeros((11, 1, 256), np.float32)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat1)
    e.set_option("loop_kernel", 3)
    for x3 in ((0,) if prec == 0 else (1, 0)):
        e.set_option("fused_x3", x3)
        lat = np.full((11, 1, 256), np.nan, np.float32)
        e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
        assert e.launch_counts()[0] == 2
        assert np.abs(lat - ref).max() < 2e-4 and np.abs(lat - lat1).max() < 2e-4, (prec, x3)
    with pytest.raises(_lib.MldHipError):
        e.set_option("fused_dbg", 1)          # the measurement builds with wrong results are not in the library (tools/loopbench)
    e.close()


@pytest.mark.parametrize("wt,groups", [(0, 4), (1, 8)])
def test_cluster_loop_sim(wt, groups):
    """loop_kernel = 4 (kernels/loop_cluster.hpp): the reverse loop as ONE launch of clusters -- 12 workgroups (3 tokens x 4 column groups) per 8 motions,
    or 24 (x 8 column groups: option cluster_groups 8, the default up to 64 motions; the members without a head enter each layer at E1) --
    that hand partial products to each other inside the launch (flags + L1-bypassing loads; the simulator runs every block of the grid as fibers
    side by side, hipsim::launch_coresident).  B = 11: two clusters, the second with 3 live motions; a 3-layer skip stack (one skip linear: the Z
    exchange), 2 steps (the end-of-step path, the ring's wrap into the next step); plain payload stores with the placement census (cluster_wt 0) and
    write-through ones (1); against the oracle and the latency family.  Launch count: condition rows + the loop."""
    dims = syn.ModelDims(num_layers=3)
    sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=12, max_frames=8, num_inference_steps=2, num_layers=3, precision=1)
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(sdv, "vae.")
    e.finalize()
    b = syn.make_batch(11, [8, 5, 3, 8, 1, 7, 2, 6, 8, 4, 8], seed=9)
    ops = O.NumpyOps(np.float32)
    ref = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, sdd), b.text_emb, b.init_latents, 7.5, 2, 4))
    e.set_option("loop_kernel", 1)
    lat1 = np.zeros((11, 1, 256), np.float32)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat1)
    e.set_option("loop_kernel", 4)
    e.set_option("cluster_wt", wt)
    e.set_option("cluster_groups", groups)
    for _ in range(2):                      # twice: every flag is back at zero when a call ends
        lat = np.full((11, 1, 256), np.nan, np.float32)
        e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
        assert e.launch_counts()[0] == 2
        assert np.abs(lat - ref).max() < 2e-4 and np.abs(lat - lat1).max() < 2e-4, (wt, groups)
    if groups == 8:
        # calls above 128 motions run several launches one after the other on shared exchange regions (ClusterArgs s_base / s_end): here 8 + 3 motions
        e.set_option("cluster_chunk", 8)
        lat = np.full((11, 1, 256), np.nan, np.float32)
        e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
        assert e.launch_counts()[0] == 3
        assert np.abs(lat - ref).max() < 2e-4 and np.abs(lat - lat1).max() < 2e-4
    # the precision mode without split arithmetic has no cluster build: refused like loop_kernel 3 where that is not built
    e.close()
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=4, max_frames=8, num_inference_steps=2, num_layers=3, precision=0)
    e.load_state_dict(sdd, "denoiser.")
    e.finalize()
    with pytest.raises(_lib.MldHipError):
        e.set_option("loop_kernel", 4)
    e.close()


def test_cluster_loop_two_skip_levels_sim():
    """A 5-layer skip stack (two input blocks, two skip connections): the cluster loop parks the rows of TWO levels and pops them in reverse order
    (cross_attention.py:48-58: xs.append in the input blocks, xs.pop() in the output blocks) -- the 3-layer models of the other simulator tests have one
    level only.  One ragged cluster (5 motions), one step, 24 workgroups; against the oracle and the launch family."""
    dims = syn.ModelDims(num_layers=5)
    sdd = syn.make_denoiser_state_dict(dims=dims)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=8, max_frames=8, num_inference_steps=1, num_layers=5, precision=1)
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(syn.make_vae_state_dict(dims=dims), "vae.")
    e.set_option("range_probe", 1)                           # finalize's probe runs the cluster loop too where the handle may pick it (probe (c), mldhip.hip)
    e.set_option("cluster_max_batch", 8)
    e.finalize()
    ns = e.numeric_status()
    assert ns["probed"] == 1 and ns["loop_split_ok"] == 1 and 0.0 <= ns["probe_err_loop"] < _lib.PROBE_TOL, ns
    b = syn.make_batch(5, [8, 5, 3, 8, 1], seed=17)
    ops = O.NumpyOps(np.float32)
    ref = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, sdd), b.text_emb, b.init_latents, 7.5, 1, 4))
    e.set_option("loop_kernel", 1)
    lat1 = np.zeros((5, 1, 256), np.float32)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat1)
    e.set_option("loop_kernel", 4)
    for groups in (8, 4):
        e.set_option("cluster_groups", groups)
        lat = np.full((5, 1, 256), np.nan, np.float32)
        e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
        assert e.launch_counts()[0] == 2
        assert np.abs(lat - ref).max() < 2e-4 and np.abs(lat - lat1).max() < 2e-4, groups
    e.close()


@pytest.mark.parametrize("groups", [8, 4])
def test_cluster_loop_bounded_waits_and_fallback_sim(groups):
    """Every wait inside the cluster launch is bounded.  Fault injection (hooks / simulator builds, option "cluster_inject"): one member never raises its first flag,
    so its partners run into the (shortened) bound -- the call must come back (no hang) with its latents poisoned, the range contract's counter must see them, and
    mldhip_numeric_status must take the handle off the cluster loop: the next call runs on another loop family and is right; loop_kernel 4 re-arms it."""
    dims = syn.ModelDims(num_layers=3)
    sdd = syn.make_denoiser_state_dict(dims=dims)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=8, max_frames=8, num_inference_steps=2, num_layers=3, precision=1)
    e.load_state_dict(sdd, "denoiser.")
    e.load_state_dict(syn.make_vae_state_dict(dims=dims), "vae.")
    e.finalize()
    b = syn.make_batch(8, [8] * 8, seed=23)
    ops = O.NumpyOps(np.float32)
    ref = np.asarray(O.diffusion_reverse(ops, O.to_backend(ops, sdd), b.text_emb, b.init_latents, 7.5, 2, 4))
    e.set_option("loop_kernel", 4)
    e.set_option("cluster_groups", groups)                   # (4: the per-wave wait of the 12-workgroup form's E2 runs into the bound too)
    e.set_option("cluster_inject", 1 + (5 if groups == 8 else 2))      # member (token 0, column group 5): a member without a head / (token 0, head 2)
    lat = np.zeros((8, 1, 256), np.float32)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
    assert e.launch_counts()[0] == 2 and np.isnan(lat).all()
    if groups == 4:
        ns = e.numeric_status()
        assert ns["nonfinite_values"] == 8 * 256 and ns["cluster_loop"] == 2, ns
    # self-healing (round 6): the kernel also set the handle's pinned host word, and every sample call reads it first -- the NEXT call is served by the launch family
    # whether or not the caller has polled mldhip_numeric_status in between (groups == 8: it has not)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)          # the handle has left the cluster loop
    assert e.launch_counts()[0] > 2 and np.abs(lat - ref).max() < 2e-4
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == (0 if groups == 4 else 8 * 256) and ns["cluster_loop"] == 2, ns
    e.set_option("cluster_inject", 0)
    e.set_option("loop_kernel", 4)                           # re-arms it
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
    assert e.launch_counts()[0] == 2 and np.abs(lat - ref).max() < 2e-4
    assert e.numeric_status()["cluster_loop"] == 1
    # entry check (round 6): a polled word that holds an epoch no fresh launch can hold (what the r05 memset-node replay fault left behind) fails the launch -- it is
    # counted and the handle falls back -- instead of being consumed as "ready"
    e.set_option("cluster_stale", 1)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
    assert np.isnan(lat).all()
    e.set_option("cluster_stale", 0)
    e.sample(b.text_emb, b.init_latents, b.lengths, latents_out=lat)
    assert e.launch_counts()[0] > 2 and np.abs(lat - ref).max() < 2e-4
    ns = e.numeric_status()
    assert ns["nonfinite_values"] == 8 * 256 and ns["cluster_loop"] == 2, ns
    e.close()


def test_sample_major_loop_is_refused_where_it_is_not_built_sim():
    """ff_size 512 has no sample-major build: loop_kernel = 3 is refused after finalize, auto never picks it."""
    dims = syn.ModelDims(num_layers=3, ff_size=512)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=4, max_frames=8, num_inference_steps=2, num_layers=3, ff_size=512)
    e.load_state_dict(syn.make_denoiser_state_dict(dims=dims), "denoiser.")
    e.finalize()
    with pytest.raises(_lib.MldHipError):
        e.set_option("loop_kernel", 3)
    e.close()
    # ... and set BEFORE finalize it is refused by finalize (round-3 advisor finding: it used to fall through to the latency chain silently)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=4, max_frames=8, num_inference_steps=2, num_layers=3, ff_size=512)
    e.load_state_dict(syn.make_denoiser_state_dict(dims=dims), "denoiser.")
    e.set_option("loop_kernel", 3)
    with pytest.raises(_lib.MldHipError):
        e.finalize()
    e.close()


def test_register_direct_ffn_kernel_sim(ow):
    """kernels/ffn_strip.hpp ("ffn_strip" = 6 / 4: 96- / 64-row strips, weights register-direct from the layer's fragment-ordered
    stream, GELU of block hb + 1 between the matrix instructions of linear2's share of block hb) against ffn_fused.hpp (= 0) and the
    oracle: decoder with ragged lengths (M = 120 frame rows: a full strip and a partial one, strips of padding skipped) and the MldVae
    encoder (S = T + 2 token rows per sample, no skipping).  Same products, another summation order: fp32 rounding apart."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=40, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    z = syn._rng(7, "g8").standard_normal((3, 1, 256)).astype(np.float32)
    g = syn._rng(3, "enc")
    lens = [40, 23, 7]
    feats_in = g.standard_normal((3, 40, 263)).astype(np.float32)
    for i, n in enumerate(lens):
        feats_in[i, n:] = 0
    eps = g.standard_normal((3, 1, 256)).astype(np.float32)
    ref = np.asarray(O.vae_decode(ops, bv, z, lens))
    outs = {}
    for opt in (0, 6, 4, 3):
        e.set_option("ffn_strip", opt)
        feats = np.full((3, 40, 263), np.nan, np.float32)
        e.vae_decode(z, lens, feats)
        mu, lv, lat = (np.zeros((3, 1, 256), np.float32) for _ in range(3))
        if opt in (0, 6, 3):               # (the 64-row strips of the encoder: covered by the decoder above and by the GPU suite)
            e.vae_encode(feats_in, lens, 40, eps, lat, mu, lv)
        assert np.isfinite(feats).all() and 1e-7 < np.abs(feats - ref).max() < 2e-4
        for i, n in enumerate(lens):
            assert np.all(feats[i, n:] == 0)
        outs[opt] = (feats, mu.copy())
    # "ffn_strip" 3 ran the decoder layers' TAIL form (out-projection + norms + feed-forward in one launch, "dec_tail" on by default):
    # the two-launch form of the same strips must agree with it to fp32 rounding
    e.set_option("ffn_strip", 3)
    e.set_option("dec_tail", 0)
    feats2 = np.full((3, 40, 263), np.nan, np.float32)
    e.vae_decode(z, lens, feats2)
    assert 0 < np.abs(feats2 - outs[3][0]).max() < 5e-5
    e.set_option("dec_tail", 1)
    for opt in (6, 4, 3):
        assert 0 < np.abs(outs[opt][0] - outs[0][0]).max() < 5e-5
        if opt != 4:
            assert 0 < np.abs(outs[opt][1] - outs[0][1]).max() < 5e-5
    with pytest.raises(_lib.MldHipError):
        e.set_option("ffn_strip", 5)
    e.close()


def test_first_decoder_layer_projects_its_input_once_sim(eng, aeng, ow, aow):
    """"dec_l0_once" (default 1): decoder layer 0's input is zeros + the positional rows (mld_vae.py:216-222, actor_vae.py:221-222) --
    the same for every sample -- so its in-projection runs over ONE sample's T rows and every (sample, head) attention workgroup reads
    those (with its own sample's length mask).  Sample 0 is the SHORTEST here: the shared projection must still cover the T rows the
    longer samples read.  Exactly the per-sample computation (same kernels, same products) in all three attention kernels, the
    TransformerDecoder stack of the action VAE included."""
    ops, _, bv = ow
    z = syn._rng(11, "l0once").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [5, 24, 17]
    ref = np.asarray(O.vae_decode(ops, bv, z, lens))

    def both_forms(e, tol):
        outs = []
        for once in (1, 0):
            e.set_option("dec_l0_once", once)
            feats = np.full((3, 24, 263), np.nan, np.float32)
            e.vae_decode(z, lens, feats)
            assert np.isfinite(feats).all() and np.abs(feats - ref).max() < tol
            for i, n in enumerate(lens):
                assert np.all(feats[i, n:] == 0)
            outs.append(feats)
        e.set_option("dec_l0_once", 1)
        assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max()
        return outs[1]

    both_forms(eng, 5e-5)                                  # exact fp32 (attn_decode_kernel)
    e = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    for fl in (0, 2):                                      # split mode: whole-K/V kernel, key-blocked kernel
        e.set_option("flash_attn", fl)
        plain = both_forms(e, 2e-4)
    with pytest.raises(_lib.MldHipError):
        e.set_option("dec_l0_once", 2)
    e.close()
    ops, _, abv = aow
    za = syn._rng(12, "l0once_a").standard_normal((3, 1, 256)).astype(np.float32)
    alens = [7, 24, 16]
    aref = np.asarray(O.actor_decode(ops, abv, za, alens))
    outs = []
    for once in (1, 0):
        aeng.set_option("dec_l0_once", once)
        feats = np.full((3, 24, 150), np.nan, np.float32)
        aeng.vae_decode(za, alens, feats)
        assert np.abs(feats - aref).max() < 5e-5
        outs.append(feats)
    aeng.set_option("dec_l0_once", 1)
    assert np.array_equal(outs[0], outs[1])


def test_key_blocked_attention_transpose_read_v_sim(ow):
    """attn_flash_x3_kernel stages V row-major like K and reads its P V fragments with ds_read_b64_tr_b16 (the simulator models the
    instruction as the guide documents it: inside a 16-lane group lane i receives column i of the [4][16] block the group's 8-byte
    reads form): ragged lengths with an odd number of key tiles and more query tiles than waves' first slots, against the oracle and
    against the whole-K/V kernel (transposed V planes, plain reads: the same products in another order)."""
    ops, _, bv = ow
    for B, T, lens in ((1, 68, [53]),):           # four key tiles = two 32-key blocks, the second one crossing the length
        e = simlib.sim_engine(max_batch=B, max_frames=T, num_inference_steps=2, precision=1)
        e.set_option("gemm_small_m", 0)
        z = syn._rng(9, "trv").standard_normal((B, 1, 256)).astype(np.float32)
        ref = np.asarray(O.vae_decode(ops, bv, z, lens))
        outs = []
        for fl in (2, 0):
            e.set_option("flash_attn", fl)
            feats = np.zeros((B, T, 263), np.float32)
            e.vae_decode(z, lens, feats)
            assert 1e-7 < np.abs(feats[:, :max(lens)] - ref).max() < 2e-4
            outs.append(feats)
        assert np.abs(outs[0] - outs[1]).max() < 5e-5
        with pytest.raises(_lib.MldHipError):
            e.set_option("attn_tr", 1)            # retired in round 4 (the transpose read is how the kernel works)
        e.close()


def test_decoder_tail_with_row_swizzled_images_sim(ow):
    """ffn_strip_x3_kernel<3, true, true> -- the one-launch decoder tail with its strip image and hidden-block image stored XOR-swizzled
    by the row (the loop's map: every write and read of an image goes through it -- the prologue's element-wise image, the
    out-projection's A fragments, the block input written by lanes in the transposed layout, the hidden blocks, the residual read back
    in the plain accumulator layout): ragged lengths, a partial last strip, against the oracle and against the two-launch form
    ("dec_tail" = 0: plain images)."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    e.set_option("ffn_strip", 3)
    z = syn._rng(13, "ffnswz").standard_normal((3, 1, 256)).astype(np.float32)
    lens = [24, 13, 7]
    ref = np.asarray(O.vae_decode(ops, bv, z, lens))
    outs = []
    for tail in (1, 0):
        e.set_option("dec_tail", tail)
        feats = np.full((3, 24, 263), np.nan, np.float32)
        e.vae_decode(z, lens, feats)
        assert np.isfinite(feats).all() and np.abs(feats - ref).max() < 2e-4
        outs.append(feats)
    assert np.abs(outs[0] - outs[1]).max() < 5e-5
    e.close()


def test_final_norm_and_linear_as_one_strip_launch_sim(ow):
    """kernels/final_strip.hpp: decoder.norm + final_layer + output[~mask.T] = 0 of MldVae.decode (mld_vae.py:240-245) as one row-strip
    launch -- rows normalised while loaded (the LayerNorm kernel's arithmetic), multiplied with the weight zero-padded to three
    128-column blocks, features written as contiguous 48-row blocks of 263-float rows -- against the oracle and against the exact-fp32
    mode (LayerNorm launch + staged GEMM): ragged lengths (M = 72 / 63 rows: a full strip and a partial one -- 15 rows x 263 floats end
    off a 16-byte boundary), a caller's output buffer that is only 4-byte aligned, one launch less than the fp32 mode's pair."""
    ops, _, bv = ow
    e = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=1)
    e.set_option("gemm_small_m", 0)
    e32 = simlib.sim_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=0)
    e32.set_option("gemm_small_m", 0)
    z = syn._rng(17, "finstrip").standard_normal((3, 1, 256)).astype(np.float32)
    first = None
    for lens in ([24, 13, 7], [21, 5, 11]):
        T = max(lens)
        ref = np.asarray(O.vae_decode(ops, bv, z, lens))
        raw = np.full(3 * T * 263 + 1, np.nan, np.float32)
        feats = raw[1:].reshape(3, T, 263)            # float-aligned only (base + 4 bytes)
        e.vae_decode(z, lens, feats)
        assert np.isnan(raw[0]) and np.isfinite(feats).all() and np.abs(feats - ref).max() < 2e-4
        for i, n in enumerate(lens):
            assert np.all(feats[i, n:] == 0)
        f32 = np.zeros((3, T, 263), np.float32)
        e32.vae_decode(z, lens, f32)
        assert np.abs(feats - f32).max() < 2e-4
        first = feats.copy()
    al = np.full((3, 21, 263), np.nan, np.float32)        # and a 16-byte aligned buffer (numpy's own allocation)
    e.vae_decode(z, [21, 5, 11], al)
    assert np.array_equal(al, first)
    e.close()
    e32.close()


def test_range_probe_keeps_or_replaces_the_split_kernels_sim():
    """The range contract of the split-f16 mode (mldhip.h) on the simulator, 3-layer models: finalize's probe ("range_probe" 1: off by
    default on the simulator, on by default on the GPU) (a) keeps both stages on the split kernels for the in-range synthetic weights,
    (b) moves the stage whose feed-forward hidden activation leaves +-65 504 to the exact-fp32 kernels -- the engine then agrees with the
    oracle on those weights, and mldhip_numeric_status says which stage fell back; the run-time counter of non-finite results reads 0
    in both cases and > 0 when the guard is switched off."""
    dims = syn.ModelDims(num_layers=3)
    ops = O.NumpyOps(np.float32)
    mean, std = syn.make_mean_std()
    b = syn.make_batch(3, [8, 5, 8], seed=5)
    for case in ("plain", "huge"):              # (the unguarded arm -- inf / saturation inside the split kernels -- is the GPU test's: tests/test_gpu_parity.py)
        sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
        if case != "plain":
            sdd["encoder.input_blocks.0.linear1.weight"] = sdd["encoder.input_blocks.0.linear1.weight"] * np.float32(2.0 ** 15)
            sdd["encoder.input_blocks.0.linear2.weight"] = sdd["encoder.input_blocks.0.linear2.weight"] * np.float32(2.0 ** -15)
        e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=8, max_frames=8, num_inference_steps=2, num_layers=3, precision=1)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae.")
        e.load_tensor("mean", mean); e.load_tensor("std", std)
        e.set_option("range_probe", 0 if case == "huge_unguarded" else 1)
        e.finalize()
        s0 = e.numeric_status()
        jr = np.asarray(O.sample(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), b.text_emb, b.init_latents, b.lengths, mean, std,
                                 steps=2))
        worst = 0.0
        for lk in (3, 1):
            e.set_option("loop_kernel", lk)
            joints = np.full((3, 8, 22, 3), np.nan, np.float32)
            e.sample(b.text_emb, b.init_latents, b.lengths, joints_out=joints)
            err = max(float(np.nan_to_num(np.abs(joints[i, :n] - jr[i, :n]), nan=np.inf).max()) for i, n in enumerate(b.lengths))
            worst = max(worst, err)
        s1 = e.numeric_status()
        e.close()
        if case == "plain":
            assert s0["probed"] == 1 and s0["loop_split_ok"] == 1 and s0["decode_split_ok"] == 1, s0
            assert 0 <= s0["probe_err_loop"] <= _lib.PROBE_TOL and 0 <= s0["probe_err_decode"] <= _lib.PROBE_TOL, s0
            assert worst < 1e-3 and s1["nonfinite_values"] == 0
        elif case == "huge":
            assert s0["probed"] == 1 and s0["loop_split_ok"] == 0 and s0["decode_split_ok"] == 1, s0      # only the denoiser was touched
            assert worst < 1e-3 and s1["nonfinite_values"] == 0, (worst, s1)
        else:
            assert s0["probed"] == 0 and s0["loop_split_ok"] == 1
            assert s1["nonfinite_values"] > 0, (worst, s1)              # inf -> NaN inside the split kernels, counted at run time
    # "range_probe" 2: weights that pass the seeded probe, a FIRST BATCH that does not (text embeddings x 3e5: the condition rows leave the half range) -- the
    # first mldhip_sample repeats the loop probe on its own batch, moves the loop to the exact-fp32 kernels and then agrees with the oracle
    sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
    e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=8, max_frames=8, num_inference_steps=2, num_layers=3, precision=1)
    e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae.")
    e.load_tensor("mean", mean); e.load_tensor("std", std)
    e.set_option("range_probe", 2)
    e.finalize()
    s0 = e.numeric_status()
    assert s0["probed"] == 1 and s0["loop_split_ok"] == 1, s0
    wild = (b.text_emb * np.float32(3e5)).astype(np.float32)
    jr = np.asarray(O.sample(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), wild, b.init_latents, b.lengths, mean, std, steps=2))
    e.set_option("loop_kernel", 1)
    joints = np.full((3, 8, 22, 3), np.nan, np.float32)
    e.sample(wild, b.init_latents, b.lengths, joints_out=joints)
    s1 = e.numeric_status()
    err = max(float(np.nan_to_num(np.abs(joints[i, :n] - jr[i, :n]), nan=np.inf).max()) for i, n in enumerate(b.lengths))
    e.close()
    assert s1["loop_split_ok"] == 0 and s1["probe_err_loop"] > _lib.PROBE_TOL, s1
    assert err < 1e-3 and s1["nonfinite_values"] == 0, (err, s1)


def test_decoder_self_attention_on_half_qkv_sim(ow, aow):
    """"dec_half" (opt-in, default 0; kernels/dec_half.hpp): the decoder's self-attention block with Q | K | V kept as one half per element -- the in-projection
    on half rows x split weights with transposed products and a half staging tile (64- and 96-row strips; ragged lengths, a partial last strip,
    all-padding strips skipped), layer 0's once-per-call projection converted by qkv_to_half_kernel, the key-blocked attention on plain half operands
    with its P V product transposed (lengths that end inside a 32-key block, more query tiles than a wave's first slot, T > 128) -- against the oracle
    (cross_attention.py:323-345, mld_vae.py:186-248) and against the fp32-Q|K|V x3 form of the same handle.  The difference between the two forms is
    the rounding of Q | K | V and of the block's input rows to half: 1e-4 .. 2e-4 on the features of these short sequences and unit-normal latents, what
    tools/precision_attribution_decoder.py's emulation of the form reads for the same case (2.1e-4) -- the reason the form is opt-in."""
    ops, _, bv = ow
    for B, T, lens in ((3, 24, [24, 13, 7]), (2, 136, [136, 71]), (1, 68, [53])):
        e = simlib.sim_engine(max_batch=B, max_frames=T, num_inference_steps=2, precision=1)
        e.set_option("gemm_small_m", 0)
        z = syn._rng(21, f"dh{B}").standard_normal((B, 1, 256)).astype(np.float32)
        ref = np.asarray(O.vae_decode(ops, bv, z, lens))
        outs = {}
        for dh in (0, 1, 6):
            e.set_option("dec_half", dh)
            feats = np.full((B, max(lens), 263), np.nan, np.float32)
            n0 = e.launch_counts()[1]
            e.vae_decode(z, lens, feats)
            outs[dh] = (feats, e.launch_counts()[1] - n0)
            assert np.isfinite(feats).all() and np.abs(feats - ref).max() < (5e-4 if dh else 2e-5), (B, T, dh, np.abs(feats - ref).max())
            for i, n in enumerate(lens):
                assert np.all(feats[i, n:] == 0)
        d = np.abs(outs[1][0] - outs[0][0]).max()
        assert 1e-6 < d < 5e-4, d
        assert np.abs(outs[6][0] - outs[1][0]).max() < 2e-5            # strip height changes the staging only
        # layer 0 of a B > 1 call pays one conversion launch more; every other layer's pair of launches is a pair again
        assert outs[1][1] == outs[0][1] + (1 if B > 1 else 0), (outs[0][1], outs[1][1])
        e.set_option("dec_l0_once", 0)                                 # layer 0 through the strip kernel like the others
        e.set_option("dec_half", 1)
        feats = np.full((B, max(lens), 263), np.nan, np.float32)
        e.vae_decode(z, lens, feats)
        assert np.abs(feats - outs[1][0]).max() < 5e-4              # (the once-per-call projection multiplies fp32 rows, the strip kernel half rows)
        assert np.abs(feats - ref).max() < 5e-4
        e.close()
    # the actor decoder (actor_vae.py:209-235) shares the layer: same option, sinusoidal queries
    opsa, _, bva = aow
    za = syn._rng(22, "dha").standard_normal((3, 1, 256)).astype(np.float32)
    alens = [24, 9, 17]
    aref = np.asarray(O.actor_decode(opsa, bva, za, alens))
    aeng = simlib.sim_action_engine(max_batch=4, max_frames=24, num_inference_steps=2, precision=1)
    aeng.set_option("gemm_small_m", 0)
    got = {}
    for dh in (0, 1):
        aeng.set_option("dec_half", dh)
        feats = np.full((3, 24, 150), np.nan, np.float32)
        aeng.vae_decode(za, alens, feats)
        assert np.abs(feats - aref).max() < 2e-4
        got[dh] = feats
    aeng.close()
    assert 0 < np.abs(got[1] - got[0]).max() < 5e-4


def test_sample_many_pipelined_requests_sim():
    """"many_pipeline" 1: mldhip_sample_many runs its requests one after the other on the single-request path (cluster loop) with two workspaces alternating -- every
    request gets what its own mldhip_sample call gives, to the bit (the simulator has no streams: this is the bookkeeping -- context rotation, per-request
    lengths / Tmax, outputs straight into each request's buffers, a request without joints); a request the cluster loop does not serve, or a handle
    with one workspace, keeps the coalesced form / is refused."""
    dims = syn.ModelDims(num_layers=3)
    sdd, sdv = syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)
    mean, std = syn.make_mean_std()

    def mk(in_flight):
        e = _lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=12, max_frames=12, num_inference_steps=2, num_layers=3, precision=1, max_in_flight=in_flight)
        e.load_state_dict(sdd, "denoiser."); e.load_state_dict(sdv, "vae."); e.load_tensor("mean", mean); e.load_tensor("std", std)
        e.finalize()
        e.set_option("cluster_max_batch", 12)
        return e
    e = mk(2)
    reqs, solo = [], []
    for seed, lens in ((1, [12, 7, 3]), (2, [9] * 8), (3, [5, 12]), (4, [11])):
        b = syn.make_batch(len(lens), lens, seed=seed)
        B, T = len(lens), max(lens)
        reqs.append(dict(text_emb=b.text_emb, init_latents=b.init_latents, lengths=lens, latents_out=np.zeros((B, 1, 256), np.float32),
                         feats_out=np.zeros((B, T, 263), np.float32), joints_out=None if seed == 3 else np.zeros((B, T, 22, 3), np.float32)))
        lat, feats, joints = np.zeros((B, 1, 256), np.float32), np.zeros((B, T, 263), np.float32), np.zeros((B, T, 22, 3), np.float32)
        e.sample(b.text_emb, b.init_latents, lens, lat, feats, joints)
        assert e.launch_counts()[0] == 2                            # condition rows + ONE cluster launch
        solo.append((lat, feats, joints))
    e.set_option("many_pipeline", 1)
    e.sample_many(reqs)
    for q, (lat, feats, joints) in zip(reqs, solo):
        assert np.array_equal(q["latents_out"], lat) and np.array_equal(q["feats_out"], feats)
        if q["joints_out"] is not None:
            assert np.array_equal(q["joints_out"], joints)
    assert e.launch_counts()[0] == 2                                # the last request's own loop, not a 24-motion chain
    assert e.numeric_status()["nonfinite_values"] == 0 and e.numeric_status()["cluster_loop"] == 1
    # a request the cluster loop does not serve (latents only: nothing to overlap) -> the coalesced chain, as before
    for q in reqs:
        q["latents_out"][:] = 0
    lat_only = [dict(text_emb=q["text_emb"], init_latents=q["init_latents"], lengths=q["lengths"], latents_out=q["latents_out"]) for q in reqs[:2]]
    e.sample_many(lat_only)
    for q, (lat, _, _) in zip(reqs[:2], solo):
        assert np.abs(q["latents_out"] - lat).max() < 2e-4
    e.close()
    e1 = mk(1)
    with pytest.raises(_lib.MldHipError):
        e1.set_option("many_pipeline", 1)
    e1.close()


def test_novae_cross_attention_folded_into_one_launch_sim(now):
    """"cross_fold" (default 1; kernels/novae.hpp cross_fold_kernel / cross2_fold_ln_kernel): LayerNorm 1 + the two-token cross-attention sub-layer + LayerNorm 2 of a
    trans_dec layer (mld_denoiser.py:208-221, cross_attention.py:323-345) as ONE launch on vectors folded from the memory tokens -- (x Wq^T + bq) . k = x . (Wq^T k) + bq . k
    and Wo (p1 v1 + p2 v2) = p1 Wo v1 + p2 Wo v2, exact algebra -- against the oracle and against the five-launch form of the same handle (query GEMM, cross2_kernel,
    out-projection GEMM, two LayerNorm passes): ragged lengths, a T that is not a multiple of the rows a workgroup serves, 4 launches less per layer."""
    ops, bd = now
    g = syn._rng(33, "crossfold")
    for R, T, lens in ((4, 30, [30, 17, 30, 9]), (2, 57, [57, 40])):
        e = simlib.sim_novae_engine(num_layers=2, max_batch=4, max_frames=64, num_inference_steps=4)
        x = g.standard_normal((R, T, 263)).astype(np.float32)
        te = (0.5 * g.standard_normal((R, 1, 768))).astype(np.float32)
        ref = np.asarray(O.denoiser_forward_novae(ops, bd, x, 321, te, lens))
        outs, launches = {}, {}
        for cf in (1, 0):
            e.set_option("cross_fold", cf)
            out = np.full((R, T, 263), np.nan, np.float32)
            n0 = e.launch_counts()[0]                                  # (this entry point does not reset the counters: differences)
            e.denoiser_forward_novae(x, 321, te, lens, T, out)
            outs[cf], launches[cf] = out, e.launch_counts()[0] - n0
            assert np.isfinite(out).all() and np.abs(out - ref).max() < 5e-5, (cf, np.abs(out - ref).max())
        assert 0 < np.abs(outs[1] - outs[0]).max() < 2e-5
        assert launches[0] - launches[1] == 4 * 2 - 1, launches       # five launches -> one, per layer; the folded form folds the text tokens once per call
        e.close()
