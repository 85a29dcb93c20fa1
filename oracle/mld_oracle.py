"""CPU oracle for the MLD sampling hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The shipped path (``mld_hip`` + ``libmldhip.so``) never does and
fails loudly when the HIP library or a GPU is missing.

What this file is
-----------------
A restatement, in plain array arithmetic, of the one path BASELINE.json's north_star
names (all citations relative to /root/reference):

  MLD.forward                mld/models/modeltype/mld.py:216-265
  MLD._diffusion_reverse     mld/models/modeltype/mld.py:290-360
  MldDenoiser.forward        mld/models/architectures/mld_denoiser.py:135-228
  Timesteps / TimestepEmbedding  mld/models/architectures/tools/embeddings.py:245-322
  SkipTransformerEncoder/Decoder mld/models/operator/cross_attention.py:18-125
  Transformer{Encoder,Decoder}Layer.forward_post  cross_attention.py:259-272, 323-345
  PositionEmbeddingLearned1D mld/models/operator/position_encoding.py:138-159
  MldVae.decode              mld/models/architectures/mld_vae.py:186-248
  MldVae.encode              mld/models/architectures/mld_vae.py:124-184   (scope row 8f.1)
  action variant (cfg 5)     mld_denoiser.py:69-77,231-279 (EmbedAction); actor_vae.py:176-235 (ActorAgnosticDecoder)
  no-VAE variant (cfg 4)     mld_denoiser.py:50-53,208-221 (pose_embd/trans_dec/pose_proj); cross_attention.py:195-233;
                             DDPM scheduler restated (diffusers absent: PARITY UNPINNED)
  lengths_to_mask            mld/utils/temos_utils.py:10-17
  feats2joints               mld/data/HumanML3D.py:41-45
  recover_from_ric           mld/data/humanml/scripts/motion_process.py:362-381,415-432
  qinv / qrot                mld/data/humanml/common/quaternion.py:16-20,54-73

and of the third-party scheduler the reference calls but does not vendor:

  diffusers.DDIMScheduler  (requirements.txt:23 "diffusers", unpinned; README badge >=0.7.2)
  call sites mld.py:81-83,310-320,345-346; params configs/modules/scheduler.yaml:1-14.

Pinning status
--------------
* Everything that lives in /root/reference is pinned: ``oracle/make_golden.py`` imports the
  reference's own ``MldDenoiser`` / ``MldVae`` / ``recover_from_ric`` in the build container,
  runs them on the seeded synthetic weights of ``mld_hip.synthetic`` and writes
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` holds this file to those vectors.
* The DDIM scheduler is **parity unpinned**: diffusers is not installed and the reference
  has no tests or golden vectors.  Its arithmetic below follows the
  published algorithm (Song et al. 2021, eq. 12; diffusers ``scheduling_ddim.py``) as restated
  in SURVEY.md App. A.3.

Two array backends share the code: ``NumpyOps`` (float32 or float64; the checker) and
``TorchOps`` (torch CPU, MKL threads; used only to time the ``cpu_baseline`` "port").
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

# ----------------------------------------------------------------------------- backends


class NumpyOps:
    name = "numpy"

    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        from scipy.special import erf  # scipy is part of the image
        self._erf = erf

    def asarray(self, x):
        return np.asarray(x, dtype=self.dtype)

    def to_numpy(self, x):
        return np.asarray(x)

    def matmul(self, a, b):
        return np.matmul(a, b)

    def swap(self, a, i, j):
        return np.swapaxes(a, i, j)

    def exp(self, a):
        return np.exp(a)

    def erf(self, a):
        return self._erf(a).astype(a.dtype)

    def sqrt(self, a):
        return np.sqrt(a)

    def cos(self, a):
        return np.cos(a)

    def sin(self, a):
        return np.sin(a)

    def relu(self, a):
        return np.maximum(a, 0)

    def sum(self, a, axis, keepdims=False):
        return a.sum(axis=axis, keepdims=keepdims)

    def amax(self, a, axis, keepdims=False):
        return a.max(axis=axis, keepdims=keepdims)

    def mean(self, a, axis, keepdims=False):
        return a.mean(axis=axis, keepdims=keepdims)

    def cat(self, xs, axis):
        return np.concatenate(xs, axis=axis)

    def stack(self, xs, axis):
        return np.stack(xs, axis=axis)

    def cumsum(self, a, axis):
        # sequential fp accumulate along axis, like torch.cumsum on CPU
        return np.cumsum(a, axis=axis, dtype=a.dtype)

    def where(self, c, a, b):
        return np.where(c, a, b)

    def zeros_like(self, a):
        return np.zeros_like(a)

    def full_like(self, a, v):
        return np.full_like(a, v)

    def mask_from_lengths(self, lengths, tmax):
        return np.arange(tmax)[None, :] < np.asarray(lengths)[:, None]


class TorchOps:
    """torch-CPU backend: same arithmetic through ATen/MKL (multi-threaded)."""
    name = "torch"

    def __init__(self, dtype="float32", device="cpu"):
        """device = "cpu" (the cpu_baseline port) or "cuda" (bench.py's eager_same_gpu comparator: the same array code through
        stock ATen on the GPU the engine runs on)."""
        import torch
        self.t = torch
        self.dtype = getattr(torch, dtype) if isinstance(dtype, str) else dtype
        self.device = torch.device(device)

    def asarray(self, x):
        t = self.t
        if isinstance(x, t.Tensor):
            return x.to(device=self.device, dtype=self.dtype)
        return t.as_tensor(np.asarray(x)).to(device=self.device, dtype=self.dtype)

    def to_numpy(self, x):
        return x.detach().cpu().numpy()

    def matmul(self, a, b):
        return self.t.matmul(a, b)

    def swap(self, a, i, j):
        return a.transpose(i, j)

    def exp(self, a):
        return self.t.exp(a)

    def erf(self, a):
        return self.t.erf(a)

    def sqrt(self, a):
        return self.t.sqrt(a)

    def cos(self, a):
        return self.t.cos(a)

    def sin(self, a):
        return self.t.sin(a)

    def relu(self, a):
        return self.t.relu(a)

    def sum(self, a, axis, keepdims=False):
        return a.sum(dim=axis, keepdim=keepdims)

    def amax(self, a, axis, keepdims=False):
        return a.amax(dim=axis, keepdim=keepdims)

    def mean(self, a, axis, keepdims=False):
        return a.mean(dim=axis, keepdim=keepdims)

    def cat(self, xs, axis):
        return self.t.cat(list(xs), dim=axis)

    def stack(self, xs, axis):
        return self.t.stack(list(xs), dim=axis)

    def cumsum(self, a, axis):
        return self.t.cumsum(a, dim=axis)

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def zeros_like(self, a):
        return self.t.zeros_like(a)

    def full_like(self, a, v):
        return self.t.full_like(a, v)

    def mask_from_lengths(self, lengths, tmax):
        t = self.t
        return t.arange(tmax, device=self.device)[None, :] < t.as_tensor(list(lengths), device=self.device)[:, None]


# ----------------------------------------------------------------------------- primitives

LN_EPS = 1e-5  # nn.LayerNorm default, cross_attention.py:248-249


def linear(ops, x, w, b=None):
    """y = x @ w.T + b  (nn.Linear; weights are [out, in])."""
    y = ops.matmul(x, ops.swap(w, -1, -2))
    return y if b is None else y + b


def layer_norm(ops, x, g, b):
    mu = ops.mean(x, -1, keepdims=True)
    xc = x - mu
    var = ops.mean(xc * xc, -1, keepdims=True)
    return xc / ops.sqrt(var + LN_EPS) * g + b


def gelu(ops, x):
    """erf GELU -- F.gelu default, cross_attention.py:408-409."""
    return 0.5 * x * (1.0 + ops.erf(x * (1.0 / math.sqrt(2.0))))


def silu(ops, x):
    return x / (1.0 + ops.exp(-x))


def softmax_last(ops, s):
    m = ops.amax(s, -1, keepdims=True)
    e = ops.exp(s - m)
    return e / ops.sum(e, -1, keepdims=True)


def mha(ops, sd, p, q_in, kv_in, nhead, key_valid=None):
    """nn.MultiheadAttention forward, batch-first restatement.

    q_in [N, L, D], kv_in [N, S, D]; packed in-proj rows 0..D-1 = q, D..2D-1 = k, 2D.. = v;
    q is pre-scaled by 1/sqrt(head_dim); key_valid [N, S] bool (False = padded key -> -inf),
    as the float -inf key_padding_mask of F.multi_head_attention_forward.
    (invoked at cross_attention.py:265-266, 332-339.)
    """
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    d = q_in.shape[-1]
    hd = d // nhead
    q = linear(ops, q_in, w[:d], b[:d])
    k = linear(ops, kv_in, w[d:2 * d], b[d:2 * d])
    v = linear(ops, kv_in, w[2 * d:], b[2 * d:])
    n, l, s = q.shape[0], q.shape[1], k.shape[1]
    q = ops.swap(q.reshape(n, l, nhead, hd), 1, 2) * (1.0 / math.sqrt(hd))
    k = ops.swap(k.reshape(n, s, nhead, hd), 1, 2)
    v = ops.swap(v.reshape(n, s, nhead, hd), 1, 2)
    sc = ops.matmul(q, ops.swap(k, -1, -2))                       # [N, H, L, S]
    if key_valid is not None:
        sc = ops.where(key_valid[:, None, None, :], sc, ops.full_like(sc, -math.inf))
    o = ops.matmul(softmax_last(ops, sc), v)                      # [N, H, L, hd]
    o = ops.swap(o, 1, 2).reshape(n, l, d)
    return linear(ops, o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def encoder_layer(ops, sd, p, x, nhead, key_valid=None):
    """TransformerEncoderLayer.forward_post (cross_attention.py:259-272), dropout off."""
    x = layer_norm(ops, x + mha(ops, sd, p + ".self_attn", x, x, nhead, key_valid),
                   sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    f = linear(ops, gelu(ops, linear(ops, x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
               sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(ops, x + f, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])


def decoder_layer(ops, sd, p, x, mem, nhead, key_valid=None):
    """TransformerDecoderLayer.forward_post (cross_attention.py:323-345), dropout off.

    The cross-attention is computed faithfully (q/k projections + softmax over the memory
    tokens); with one memory token its softmax is identically 1.
    """
    x = layer_norm(ops, x + mha(ops, sd, p + ".self_attn", x, x, nhead, key_valid),
                   sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    x = layer_norm(ops, x + mha(ops, sd, p + ".multihead_attn", x, mem, nhead),
                   sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    f = linear(ops, gelu(ops, linear(ops, x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
               sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(ops, x + f, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])


def _num_block(sd, p):
    n = 0
    while f"{p}.linear_blocks.{n}.weight" in sd:
        n += 1
    return n


def skip_transformer(ops, sd, p, x, nhead, layer_fn):
    """SkipTransformerEncoder/Decoder.forward (cross_attention.py:41-64, 89-125).

    ``layer_fn(prefix, x)`` applies one layer; U-Net style: push after each input block, pop
    (LIFO) and ``Linear(cat[x, skip])`` before each output block, final LayerNorm.
    """
    nb = _num_block(sd, p)
    xs = []
    for i in range(nb):
        x = layer_fn(f"{p}.input_blocks.{i}", x)
        xs.append(x)
    x = layer_fn(f"{p}.middle_block", x)
    for i in range(nb):
        x = ops.cat([x, xs.pop()], -1)
        x = linear(ops, x, sd[f"{p}.linear_blocks.{i}.weight"], sd[f"{p}.linear_blocks.{i}.bias"])
        x = layer_fn(f"{p}.output_blocks.{i}", x)
    return layer_norm(ops, x, sd[p + ".norm.weight"], sd[p + ".norm.bias"])


# ----------------------------------------------------------------------------- denoiser


def timestep_embedding(ops, t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0):
    """get_timestep_embedding (embeddings.py:245-285): [cos(t f) | sin(t f)] when flipped.

    The frequency table is built in float32 exactly as the reference does
    (exp of a float32 arange scaled in float32), then cast to the backend dtype.
    """
    half = dim // 2
    expo = (-math.log(max_period) * np.arange(half, dtype=np.float32)) / np.float32(half - freq_shift)
    freqs = ops.asarray(np.exp(expo.astype(np.float32)))
    ang = ops.asarray(np.asarray(t, dtype=np.float32).reshape(-1, 1)) * freqs[None, :]
    emb = ops.cat([ops.sin(ang), ops.cos(ang)], -1)
    if flip_sin_to_cos:
        emb = ops.cat([emb[:, half:], emb[:, :half]], -1)
    return emb


def denoiser_forward_action(ops, sd, sample, timestep, actions, nhead=4, guidance_scale=7.5):
    """MldDenoiser.forward, action condition (mld_denoiser.py:69-77,135-228) with EmbedAction in eval mode
    (mld_denoiser.py:249-260): rows of the embedding table, the FIRST half of the batch replaced by zeros when
    guidance_scale > 1 (the unconditional half of the CFG batch).  sample [R,1,D], actions int [R] -> [R,1,D]."""
    d = sample.shape[-1]
    temb0 = timestep_embedding(ops, [float(timestep)], d)
    temb = linear(ops, silu(ops, linear(ops, temb0, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                  sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    temb = temb[None, :, :] + ops.zeros_like(sample)
    idx = [int(a) for a in np.asarray(actions).reshape(-1)]
    emb = ops.stack([sd["emb_proj.action_embedding"][i] for i in idx], 0)[:, None, :]            # [R,1,D]
    if guidance_scale > 1.0:
        half = len(idx) // 2
        emb = ops.cat([ops.zeros_like(emb[:half]), emb[half:]], 0)
    xseq = ops.cat([sample, temb, emb], 1)
    xseq = xseq + ops.swap(sd["query_pos.pe"][: xseq.shape[1]], 0, 1)
    out = skip_transformer(ops, sd, "encoder", xseq, nhead, lambda p, x: encoder_layer(ops, sd, p, x, nhead))
    return out[:, : sample.shape[1], :]


def denoiser_forward(ops, sd, sample, timestep, text_emb, nhead=4):
    """MldDenoiser.forward, text condition / trans_enc / skip / learned PE
    (mld_denoiser.py:135-228).  sample [R,1,D], timestep scalar, text_emb [R,1,768] -> [R,1,D].
    """
    r = sample.shape[0]
    temb0 = timestep_embedding(ops, [float(timestep)], sd["time_embedding.linear_1.weight"].shape[1])
    temb = linear(ops, silu(ops, linear(ops, temb0, sd["time_embedding.linear_1.weight"],
                                        sd["time_embedding.linear_1.bias"])),
                  sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])   # [1, D]
    temb = temb[None, :, :] + ops.zeros_like(sample)                                           # expand to [R,1,D]
    cemb = linear(ops, ops.relu(text_emb), sd["emb_proj.1.weight"], sd["emb_proj.1.bias"])    # ReLU quirk :65-68
    xseq = ops.cat([sample, temb, cemb], 1)                                                    # tokens [lat,time,text] :187
    xseq = xseq + ops.swap(sd["query_pos.pe"][: xseq.shape[1]], 0, 1)                          # learned PE :196
    out = skip_transformer(ops, sd, "encoder", xseq, nhead,
                           lambda p, x: encoder_layer(ops, sd, p, x, nhead))
    return out[:, : sample.shape[1], :]                                                        # token 0 :206


# ----------------------------------------------------------------------------- scheduler (third party, restated)


def denoiser_forward_novae(ops, sd, sample, timestep, text_emb, lengths: Sequence[int], nhead=4):
    """MldDenoiser.forward, diffusion-only (VAE_TYPE 'no') + arch trans_dec (mld_denoiser.py:50-53,144-146,208-221):
    pose_embd -> + query_pos over the T frames; memory = [time, text] tokens + mem_pos; TransformerDecoder
    (cross_attention.py:195-233: plain layer stack + final norm; self-attention sees ALL T frames, no mask);
    pose_proj; rows t >= len zeroed.  sample [R,T,nfeats], text_emb [R,1,text_dim] -> [R,T,nfeats]."""
    r, t = sample.shape[0], sample.shape[1]
    d = sd["pose_embd.weight"].shape[0]
    temb0 = timestep_embedding(ops, [float(timestep)], sd["time_embedding.linear_1.weight"].shape[1])
    temb = linear(ops, silu(ops, linear(ops, temb0, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                  sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])          # [1, d]
    cemb = linear(ops, ops.relu(text_emb), sd["emb_proj.1.weight"], sd["emb_proj.1.bias"])          # [R, 1, d]
    mem = ops.cat([temb[None, :, :] + ops.zeros_like(cemb), cemb], 1)                                # [R, 2, d] time first
    mem = mem + ops.swap(sd["mem_pos.pe"][:2], 0, 1)
    x = linear(ops, sample, sd["pose_embd.weight"], sd["pose_embd.bias"]) + ops.swap(sd["query_pos.pe"][:t], 0, 1)
    i = 0
    while f"decoder.layers.{i}.linear1.weight" in sd:
        x = decoder_layer(ops, sd, f"decoder.layers.{i}", x, mem, nhead, None)
        i += 1
    x = layer_norm(ops, x, sd["decoder.norm.weight"], sd["decoder.norm.bias"])
    y = linear(ops, x, sd["pose_proj.weight"], sd["pose_proj.bias"])
    valid = ops.mask_from_lengths(lengths, t)
    assert d == x.shape[-1]
    return ops.where(valid[:, :, None], y, ops.zeros_like(y))


class DDPMSchedule:
    """diffusers.DDPMScheduler as the reference configures it (configs/modules_novae/scheduler.yaml:16-29: 1000 train
    steps, scaled_linear betas, variance_type fixed_small, clip_sample false, epsilon prediction), restated from the
    published algorithm.  THIRD PARTY, ABSENT HERE: PARITY UNPINNED.  float32 tables.  At step
    ratio 1 (the only shipped setting: 1000 inference steps) alpha_t / beta_t are read from the tables, as the diffusers
    releases contemporary with the reference do; for other ratios (test-size runs) they are recomputed from the cumulative
    products (alpha_t = ab_t / ab_prev, as newer diffusers do) -- mathematically equal at ratio 1."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        f = np.float32
        self.n_train = num_train_timesteps
        self.betas = (np.linspace(f(beta_start) ** f(0.5), f(beta_end) ** f(0.5), num_train_timesteps, dtype=f) ** 2).astype(f)
        self.alphas = (f(1.0) - self.betas).astype(f)
        self.alphas_cumprod = np.cumprod(self.alphas, dtype=f)
        self.init_noise_sigma = 1.0
        self.ratio = 1

    def set_timesteps(self, n):
        if self.n_train % n:
            raise ValueError("num_train_timesteps must be a multiple of num_inference_steps")
        self.ratio = self.n_train // n
        return (np.arange(0, n) * self.ratio)[::-1].astype(np.int64)

    def coeffs(self, t):
        """(sqrt(ab_t), sqrt(1-ab_t), c_x0, c_x, sigma) of one step, float32."""
        f = np.float32
        prev = t - self.ratio
        ab_t = self.alphas_cumprod[t]
        ab_p = self.alphas_cumprod[prev] if prev >= 0 else f(1.0)
        if self.ratio == 1:                      # the tables, as the diffusers releases of the reference's time read them
            a_t, b_t = self.alphas[t], self.betas[t]
        else:                                    # any other ratio: alpha_t = ab_t / ab_prev (newer diffusers)
            a_t = f(ab_t / ab_p)
            b_t = f(f(1.0) - a_t)
        bp_t, bp_p = f(f(1.0) - ab_t), f(f(1.0) - ab_p)
        c_x0 = f(np.sqrt(ab_p, dtype=f) * b_t / bp_t)
        c_x = f(np.sqrt(a_t, dtype=f) * bp_p / bp_t)
        var = f(max(float(bp_p / bp_t * b_t), 1e-20))
        sigma = f(np.sqrt(var, dtype=f)) if t > 0 else f(0.0)
        return np.sqrt(ab_t, dtype=f), np.sqrt(bp_t, dtype=f), c_x0, c_x, sigma

    def step(self, eps, t, x, noise=None):
        sa, sb, c0, c1, sg = (float(v) for v in self.coeffs(int(t)))
        x0 = (x - sb * eps) / sa
        y = c0 * x0 + c1 * x
        if sg != 0.0:
            y = y + sg * noise
        return y


def sample_novae(ops, sd_den, text_emb, init_latents, lengths, step_noise, mean=None, std=None, guidance_scale=7.5,
                 steps=1000, nhead=4):
    """MLD.forward with vae_type 'no' (mld.py:216-265,290-360): latents are raw motion [B,T,nfeats]; DDPM ancestral
    sampling with the per-step Gaussian draws injected (step_noise [steps,B,T,nfeats]; the reference draws them from
    torch's global generator inside scheduler.step); "decode" is the identity.  Returns feats (and joints if mean/std)."""
    sch = DDPMSchedule()
    lat = init_latents * sch.init_noise_sigma
    b = lat.shape[0]
    for i, t in enumerate(sch.set_timesteps(steps)):
        eps = denoiser_forward_novae(ops, sd_den, ops.cat([lat, lat], 0), t, text_emb, list(lengths) * 2, nhead)
        u, c = eps[:b], eps[b:]
        lat = sch.step(u + guidance_scale * (c - u), int(t), lat, step_noise[i])
    if mean is None:
        return lat
    return feats2joints(ops, lat, mean, std), lat


PHILOX_M0, PHILOX_M1, PHILOX_W0, PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox_normal(n, seed, step):
    """The engine's counter-based N(0,1) stream (kernels/novae.hpp: Philox4x32-10 + Box-Muller), numpy restatement:
    element e belongs to counter (e // 4, step), key = seed; uniform = (x >> 8 + 0.5) / 2^24."""
    nq = (n + 3) // 4
    c = [np.arange(nq, dtype=np.uint64) & 0xFFFFFFFF, np.arange(nq, dtype=np.uint64) >> 32,
         np.full(nq, step, np.uint64), np.zeros(nq, np.uint64)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = PHILOX_M0 * c[0], PHILOX_M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + PHILOX_W0) & 0xFFFFFFFF, (k1 + PHILOX_W1) & 0xFFFFFFFF
    f = np.float32
    u = [((x >> 8).astype(f) * f(1.0 / 16777216.0) + f(0.5 / 16777216.0)).astype(f) for x in c]
    r0, t0 = np.sqrt(f(-2.0) * np.log(u[0])), f(2 * np.pi) * u[1]
    r1, t1 = np.sqrt(f(-2.0) * np.log(u[2])), f(2 * np.pi) * u[3]
    z = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], 1).astype(f)
    return z.reshape(-1)[:n]


class DDIMSchedule:
    """diffusers.DDIMScheduler as configured by configs/modules/scheduler.yaml:1-14.

    scaled_linear betas, float32 tables, steps_offset=1, set_alpha_to_one=False,
    clip_sample=False, eta=0, prediction_type epsilon.  PARITY UNPINNED (see module header).
    """

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False):
        betas = np.linspace(np.float32(beta_start) ** np.float32(0.5), np.float32(beta_end) ** np.float32(0.5),
                            num_train_timesteps, dtype=np.float32) ** 2
        self.betas = betas.astype(np.float32)
        self.alphas_cumprod = np.cumprod((np.float32(1.0) - self.betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = (np.arange(0, n) * ratio).round()[::-1].astype(np.int64) + self.steps_offset
        return self.timesteps

    def coeffs(self, t):
        """(sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)) in float32."""
        prev = int(t) - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[int(t)]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        one = np.float32(1.0)
        return (np.sqrt(a_t, dtype=np.float32), np.sqrt(one - a_t, dtype=np.float32),
                np.sqrt(a_p, dtype=np.float32), np.sqrt(one - a_p, dtype=np.float32))

    def step(self, eps, t, x):
        """x0 = (x - sqrt(1-abar_t) eps)/sqrt(abar_t);  x' = sqrt(abar_p) x0 + sqrt(1-abar_p) eps."""
        sa, sb, pa, pb = self.coeffs(t)
        x0 = (x - float(sb) * eps) / float(sa)
        return float(pa) * x0 + float(pb) * eps


def diffusion_reverse(ops, sd_den, text_emb, init_latents, guidance_scale=7.5, steps=50, nhead=4,
                      schedule: Optional[DDIMSchedule] = None, trace: Optional[list] = None):
    """MLD._diffusion_reverse (mld.py:290-360) with injected start noise.

    text_emb [2B,1,768] (uncond half first, mld.py:224-231), init_latents [B,1,D] -> [B,1,D].
    """
    sch = schedule or DDIMSchedule()
    lat = init_latents * sch.init_noise_sigma
    for t in sch.set_timesteps(steps):
        x = ops.cat([lat, lat], 0)                                         # CFG duplicate :325
        eps = denoiser_forward(ops, sd_den, x, t, text_emb, nhead)
        b = lat.shape[0]
        u, c = eps[:b], eps[b:]
        eps = u + guidance_scale * (c - u)                                 # :339-342
        lat = sch.step(eps, t, lat)                                        # :345-346
        if trace is not None:
            trace.append(ops.to_numpy(lat).copy())
    return lat


# ----------------------------------------------------------------------------- VAE decode + joints


def vae_decode(ops, sd, z, lengths: Sequence[int], nhead=4):
    """MldVae.decode, arch encoder_decoder / PE mld (mld_vae.py:186-248).

    z [B, 1, D] (batch-first view of the reference's [1,B,D]) -> feats [B, max(lengths), nfeats],
    zeros at padded frames.
    """
    b, tmax = len(lengths), int(max(lengths))
    valid = ops.mask_from_lengths(lengths, tmax)                                         # lengths_to_mask
    pe = ops.swap(sd["query_pos_decoder.pe"][:tmax], 0, 1)                               # [1,T,D]
    q = pe + ops.zeros_like(z[:, :1, :])                                                 # zeros + PE, :190,224
    out = skip_transformer(ops, sd, "decoder", q, nhead,
                           lambda p, x: decoder_layer(ops, sd, p, x, z, nhead, valid))
    feats = linear(ops, out, sd["final_layer.weight"], sd["final_layer.bias"])
    return ops.where(valid[:, :, None], feats, ops.zeros_like(feats))                    # :245


def actor_decode(ops, sd, z, lengths: Sequence[int], nhead=4):
    """ActorVae.decode -> ActorAgnosticDecoder.forward (actor_vae.py:209-235): sinusoidal-PE time queries,
    stock post-norm nn.TransformerDecoderLayer stack (same sub-layer order as cross_attention.py:323-345), NO
    final LayerNorm, final_layer, zero padded frames.  z [B,1,D] -> feats [B, max(lengths), nfeats]."""
    b, tmax = len(lengths), int(max(lengths))
    valid = ops.mask_from_lengths(lengths, tmax)
    x = ops.swap(sd["decoder.sequence_pos_encoding.pe"][:tmax], 0, 1) + ops.zeros_like(z[:, :1, :])
    i = 0
    while f"decoder.seqTransDecoder.layers.{i}.linear1.weight" in sd:
        x = decoder_layer(ops, sd, f"decoder.seqTransDecoder.layers.{i}", x, z, nhead, valid)
        i += 1
    feats = linear(ops, x, sd["decoder.final_layer.weight"], sd["decoder.final_layer.bias"])
    return ops.where(valid[:, :, None], feats, ops.zeros_like(feats))


def actor_encode(ops, sd, feats, lengths: Sequence[int], eps=None, nhead=4):
    """ActorVae.encode -> ActorAgnosticEncoder.forward (actor_vae.py:64-76,121-175): skel_embedding, [mu_token,
    logvar_token, frames] + sinusoidal PE, stock post-norm nn.TransformerEncoderLayer stack with the key-padding mask
    (tokens always visible), NO final norm; mu = out[0], logvar = out[1].  feats [B,T,nfeats] ->
    (latent [B,1,D] or None, mu [B,1,D], logvar [B,1,D])."""
    b, t = feats.shape[0], feats.shape[1]
    x = linear(ops, feats, sd["encoder.skel_embedding.weight"], sd["encoder.skel_embedding.bias"])
    tok = ops.stack([sd["encoder.mu_token"], sd["encoder.logvar_token"]], 0)[None, :, :] + ops.zeros_like(x[:, :2, :])
    xseq = ops.cat([tok, x], 1) + ops.swap(sd["encoder.sequence_pos_encoding.pe"][: t + 2], 0, 1)
    valid = ops.mask_from_lengths([n + 2 for n in lengths], t + 2)
    i = 0
    while f"encoder.seqTransEncoder.layers.{i}.linear1.weight" in sd:
        xseq = encoder_layer(ops, sd, f"encoder.seqTransEncoder.layers.{i}", xseq, nhead, valid)
        i += 1
    mu, logvar = xseq[:, 0:1, :], xseq[:, 1:2, :]
    latent = None if eps is None else mu + ops.sqrt(ops.exp(logvar)) * eps
    return latent, mu, logvar


def sample_action(ops, sd_den, sd_vae, actions, init_latents, lengths, guidance_scale=7.5, steps=50, nhead=4,
                  return_intermediates=False):
    """MLD.a2m_eval's sampling core (mld.py:710-735): cond = cat(zeros_like(actions), actions), reverse diffusion,
    ActorVae decode.  Returns feats [B, T, nfeats] (joints need SMPL, unavailable: compared at feature level)."""
    sch = DDIMSchedule()
    lat = init_latents * sch.init_noise_sigma
    acts = np.asarray(actions).reshape(-1)
    cond = np.concatenate([np.zeros_like(acts), acts])
    for t in sch.set_timesteps(steps):
        eps = denoiser_forward_action(ops, sd_den, ops.cat([lat, lat], 0), t, cond, nhead, guidance_scale)
        b = lat.shape[0]
        u, c = eps[:b], eps[b:]
        lat = sch.step(u + guidance_scale * (c - u), t, lat)
    feats = actor_decode(ops, sd_vae, lat, lengths, nhead)
    return (feats, lat) if return_intermediates else feats


def vae_encode(ops, sd, feats, lengths: Sequence[int], eps=None, nhead=4):
    """MldVae.encode, PE mld / MLP_DIST false (mld_vae.py:124-184).

    feats [B, T, nfeats] (zero padded), lengths -> (latent [B,1,D], mu [B,1,D], logvar [B,1,D]).
    Tokens: [mu token, logvar token, frame 0 .. frame T-1] (global_motion_token rows first, :143-154), learned
    PE over the T+2 positions, SkipTransformerEncoder with key-padding mask (tokens always valid), first two
    output rows are mu / logvar; rsample = mu + exp(logvar)^0.5 * eps with the N(0,1) draw injected.
    """
    b, t = feats.shape[0], feats.shape[1]
    valid = ops.mask_from_lengths([int(x) + 2 for x in lengths], t + 2)            # aug_mask: 2 tokens + frames
    x = linear(ops, feats, sd["skel_embedding.weight"], sd["skel_embedding.bias"])   # [B,T,D]
    tok = sd["global_motion_token"][None, :, :] + ops.zeros_like(x[:, :1, :])        # [B,2,D] (tile)
    xseq = ops.cat([tok, x], 1)
    xseq = xseq + ops.swap(sd["query_pos_encoder.pe"][: t + 2], 0, 1)
    out = skip_transformer(ops, sd, "encoder", xseq, nhead,
                           lambda p, h: encoder_layer(ops, sd, p, h, nhead, valid))
    mu, logvar = out[:, 0:1, :], out[:, 1:2, :]
    if eps is None:
        return mu, mu, logvar
    std = ops.sqrt(ops.exp(logvar))                                                  # logvar.exp().pow(0.5)
    return mu + std * eps, mu, logvar


def feats2joints(ops, feats, mean, std, njoints=22):
    """HumanML3DDataModule.feats2joints + recover_from_ric (HumanML3D.py:41-45,
    motion_process.py:362-381,415-432, quaternion.py:16-20,54-73).  [B,T,263] -> [B,T,22,3].
    """
    f = feats * std + mean
    rot_vel = f[..., 0]
    ang = ops.cat([ops.zeros_like(rot_vel[..., :1]), rot_vel[..., :-1]], -1)
    ang = ops.cumsum(ang, -1)                                          # yaw_t = sum_{s<t} f[s,0]
    c, s = ops.cos(ang), ops.sin(ang)

    def rot(vx, vy, vz):
        # qrot(qinv((c,0,s,0)), v) = v + 2 (w (u x v) + u x (u x v)),  u = (0,-s,0), w = c
        uvx, uvy, uvz = -s * vz, ops.zeros_like(vx), s * vx            # u x v
        uuvx, uuvy, uuvz = -s * uvz, ops.zeros_like(vx), s * uvx       # u x (u x v)
        return (vx + 2 * (c * uvx + uuvx), vy + 2 * (c * uvy + uuvy), vz + 2 * (c * uvz + uuvz))

    zero = ops.zeros_like(rot_vel[..., :1])
    vx = ops.cat([zero, f[..., :-1, 1]], -1)                           # r_pos[1:, [0,2]] = data[:-1, 1:3]
    vz = ops.cat([zero, f[..., :-1, 2]], -1)
    dx, _, dz = rot(vx, ops.zeros_like(vx), vz)
    rx, rz = ops.cumsum(dx, -1), ops.cumsum(dz, -1)
    ry = f[..., 3]
    ric = f[..., 4:4 + (njoints - 1) * 3]
    ric = ric.reshape(tuple(ric.shape[:-1]) + (njoints - 1, 3))
    cj, sj = c[..., None], s[..., None]
    px, py, pz = ric[..., 0], ric[..., 1], ric[..., 2]
    uvx, uvz = -sj * pz, sj * px
    uuvx, uuvz = -sj * uvz, sj * uvx
    jx = px + 2 * (cj * uvx + uuvx) + rx[..., None]
    jy = py
    jz = pz + 2 * (cj * uvz + uuvz) + rz[..., None]
    joints = ops.stack([jx, jy, jz], -1)
    root = ops.stack([rx, ry, rz], -1)[..., None, :]
    return ops.cat([root, joints], -2)


def sample(ops, sd_den, sd_vae, text_emb, init_latents, lengths, mean, std,
           guidance_scale=7.5, steps=50, nhead=4, return_intermediates=False):
    """MLD.forward (mld.py:216-265) after the text encoder: reverse diffusion -> decode -> joints.

    Returns joints [B, Tmax, 22, 3] (padded; callers slice [:len_i] = remove_padding, temos_utils.py:24-28).
    """
    lat = diffusion_reverse(ops, sd_den, text_emb, init_latents, guidance_scale, steps, nhead)
    feats = vae_decode(ops, sd_vae, lat, lengths, nhead)
    joints = feats2joints(ops, feats, mean, std)
    if return_intermediates:
        return joints, feats, lat
    return joints


def to_backend(ops, sd: Dict[str, np.ndarray]):
    return {k: ops.asarray(v) for k, v in sd.items()}
