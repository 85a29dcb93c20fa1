"""Closed-form checks of the DDIM / DDPM restatements that do NOT go through oracle/ code.

The schedulers are third-party arithmetic (diffusers, absent offline: "parity unpinned"), and every
end-to-end fixture uses the oracle's scheduler on both sides, so a mistake shared by oracle, host mirror and engine
would be invisible there.  These tests pin all three implementations to identities derived straight from the papers
(Song et al. 2021 eq. 12 with sigma = 0; Ho et al. 2020 eq. 6-7) evaluated in float64 from the beta schedule alone:

  * DDIM, eta = 0: if x_t = sqrt(abar_t) x0 + sqrt(1 - abar_t) eps and the model returns exactly that eps, one step gives
    x_prev = sqrt(abar_prev) x0 + sqrt(1 - abar_prev) eps and pred_original_sample = x0 -- at every step of the 50-step
    grid, so the whole trajectory is known in closed form.
  * DDPM: prev = mu_tilde(x_t, x0) + sqrt(beta_tilde) z with Ho et al.'s posterior coefficients; at t = 0 no noise.
  * the timestep grids: DDIM leading spacing with steps_offset 1 -> 981, 961, ..., 1; DDPM 999, ..., 0.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import simlib  # noqa: E402
from mld_hip.scheduler import HipDDIMScheduler, HipDDPMScheduler  # noqa: E402
from oracle import mld_oracle as O  # noqa: E402

N, BETA0, BETA1 = 1000, 0.00085, 0.012


def abar64():
    """scaled_linear schedule in float64, independent of every implementation under test."""
    betas = np.linspace(BETA0 ** 0.5, BETA1 ** 0.5, N, dtype=np.float64) ** 2
    return betas, np.cumprod(1.0 - betas)


def ddim_prev64(ab, t, ratio=20):
    return ab[t - ratio] if t - ratio >= 0 else ab[0]            # set_alpha_to_one = False: final_alpha_cumprod = abar[0]


@pytest.fixture(scope="module")
def engine():
    e = simlib.sim_engine(max_batch=2, max_frames=16)
    yield e
    e.close()


def test_timestep_grids(engine):
    want = np.arange(49, -1, -1) * 20 + 1
    assert want[0] == 981 and want[-1] == 1 and len(want) == 50
    s = HipDDIMScheduler(num_train_timesteps=N, beta_start=BETA0, beta_end=BETA1, beta_schedule="scaled_linear", clip_sample=False,
                         set_alpha_to_one=False, steps_offset=1)
    s.set_timesteps(50)
    np.testing.assert_array_equal(s.timesteps.numpy(), want)
    np.testing.assert_array_equal(O.DDIMSchedule().set_timesteps(50), want)
    np.testing.assert_array_equal(engine.timesteps(), want)
    p = HipDDPMScheduler(num_train_timesteps=N, beta_start=BETA0, beta_end=BETA1, beta_schedule="scaled_linear", clip_sample=False)
    p.set_timesteps(1000)
    np.testing.assert_array_equal(p.timesteps.numpy(), np.arange(999, -1, -1))
    np.testing.assert_array_equal(O.DDPMSchedule().set_timesteps(1000), np.arange(999, -1, -1))


def test_alphas_cumprod_tables_against_float64(engine):
    _, ab = abar64()
    s = HipDDIMScheduler(num_train_timesteps=N, beta_start=BETA0, beta_end=BETA1, beta_schedule="scaled_linear", clip_sample=False,
                         set_alpha_to_one=False, steps_offset=1)
    for name, tab in (("host mirror", s.alphas_cumprod.numpy()), ("oracle", O.DDIMSchedule().alphas_cumprod),
                      ("engine", engine.alphas_cumprod())):
        rel = np.abs(tab.astype(np.float64) - ab) / ab
        assert rel.max() < 2e-4, (name, rel.max())              # float32 cumprod of 1000 factors: ~1e-4 relative at the tail
        assert tab[0] == pytest.approx(1.0 - BETA0, rel=1e-6) and np.all(np.diff(tab) < 0)
    assert ab[-1] == pytest.approx(0.0047, rel=0.02)              # the Stable-Diffusion schedule's terminal signal level


def test_ddim_eta0_recovers_x0_and_follows_the_closed_form_trajectory(engine):
    _, ab = abar64()
    rng = np.random.default_rng(7)
    x0 = rng.standard_normal((2, 1, 256))
    eps = rng.standard_normal((2, 1, 256))
    steps = np.arange(49, -1, -1) * 20 + 1
    s = HipDDIMScheduler(num_train_timesteps=N, beta_start=BETA0, beta_end=BETA1, beta_schedule="scaled_linear", clip_sample=False,
                         set_alpha_to_one=False, steps_offset=1)
    s.set_timesteps(50)
    o = O.DDIMSchedule()
    o.set_timesteps(50)
    x_t = (np.sqrt(ab[981]) * x0 + np.sqrt(1 - ab[981]) * eps)
    xs = {"host": torch.from_numpy(x_t.astype(np.float32)), "oracle": x_t.astype(np.float32), "engine": x_t.astype(np.float32)}
    e32 = eps.astype(np.float32)
    for t in steps:
        ap = ddim_prev64(ab, int(t))
        want = np.sqrt(ap) * x0 + np.sqrt(1 - ap) * eps           # closed form: same x0, same eps, one noise level down
        out = s.step(torch.from_numpy(e32), int(t), xs["host"], eta=0.0)
        assert np.abs(out.pred_original_sample.numpy() - x0).max() < 2e-4 / np.sqrt(ab[int(t)])   # x0 recovered (fp32, /sqrt(abar_t))
        xs["host"] = out.prev_sample
        xs["oracle"] = o.step(e32, int(t), xs["oracle"]).astype(np.float32)
        nxt = np.empty_like(xs["engine"])
        engine.ddim_step(e32, int(t), xs["engine"], nxt, nxt.size)                     # C ABI mldhip_ddim_step (simulator build)
        xs["engine"] = nxt
        for k in ("host", "oracle", "engine"):
            got = xs[k].numpy() if k == "host" else xs[k]
            assert np.abs(got - want).max() < 5e-4, (k, int(t), np.abs(got - want).max())
    # after the last step (t = 1 -> "prev" = final_alpha_cumprod = abar[0]) the sample sits at noise level abar[0]
    assert np.abs(xs["engine"] - (np.sqrt(ab[0]) * x0 + np.sqrt(1 - ab[0]) * eps)).max() < 5e-4


def test_ddpm_step_is_the_ho_posterior():
    betas, ab = abar64()
    rng = np.random.default_rng(11)
    x0, eps, z = (rng.standard_normal((2, 5, 263)) for _ in range(3))
    p = HipDDPMScheduler(num_train_timesteps=N, beta_start=BETA0, beta_end=BETA1, beta_schedule="scaled_linear", clip_sample=False)
    p.set_timesteps(1000)
    o = O.DDPMSchedule()
    o.set_timesteps(1000)
    L = simlib._lib
    eng = L.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=8, latent_dim=512, vae_arch=L.VAE_NONE,
                   denoiser_arch=L.ARCH_TRANS_DEC, scheduler_type=L.SCHED_DDPM, num_inference_steps=1000, steps_offset=0)
    for t in (999, 742, 500, 20, 1, 0):
        ab_t, ab_p = ab[t], (ab[t - 1] if t > 0 else 1.0)
        x_t = np.sqrt(ab_t) * x0 + np.sqrt(1 - ab_t) * eps
        mu = np.sqrt(ab_p) * betas[t] / (1 - ab_t) * x0 + np.sqrt(1 - betas[t]) * (1 - ab_p) / (1 - ab_t) * x_t      # Ho et al. eq. 7
        var = (1 - ab_p) / (1 - ab_t) * betas[t]                                                                     # beta_tilde
        want = mu + (np.sqrt(var) * z if t > 0 else 0.0)
        f = lambda a: a.astype(np.float32)
        got_h = p.step(torch.from_numpy(f(eps)), t, torch.from_numpy(f(x_t)), noise=torch.from_numpy(f(z))).prev_sample.numpy()
        got_o = o.step(f(eps), t, f(x_t), f(z))
        tol = 3e-4 / np.sqrt(ab_t)                      # x0 is reconstructed by dividing by sqrt(abar_t): fp32 noise scales with it
        got_e = np.empty_like(got_o)
        eng.ddpm_step(f(eps), t, f(x_t), f(z), got_e, got_e.size)                     # C ABI mldhip_ddpm_step (simulator build)
        assert np.abs(got_h - want).max() < tol and np.abs(got_o - want).max() < tol, (t, np.abs(got_h - want).max())
        assert np.abs(got_e - want).max() < tol, (t, np.abs(got_e - want).max())
        sa, sb, c0, c1, sg = p.coeffs(t)
        assert sg == pytest.approx(np.sqrt(var) if t > 0 else 0.0, rel=2e-4, abs=1e-12)
        assert c0 == pytest.approx(np.sqrt(ab_p) * betas[t] / (1 - ab_t), rel=3e-4)
        assert c1 == pytest.approx(np.sqrt(1 - betas[t]) * (1 - ab_p) / (1 - ab_t), rel=3e-4)
    eng.close()
