"""Deterministic synthetic weights / inputs for the MLD sampling hot path.

No checkpoint, CLIP weights or HumanML3D statistics are reachable offline, so every
test and benchmark in this repo runs on *synthetic* tensors produced here.  The
generator is pure numpy (PCG64 streams keyed by tensor name) so the very same bytes
are produced in the build container and on the GPU box.

Key names and shapes follow the reference checkpoint contract:
  denoiser  -> mld/models/architectures/mld_denoiser.py:40-133  (126 tensors, 8 349 184 params)
  vae       -> mld/models/architectures/mld_vae.py:35-112       (297 tensors, 18 032 135 params)
Lightning checkpoints prefix them with ``denoiser.`` / ``vae.`` (mld/models/modeltype/base.py:96-127).

Init statistics mimic the reference's (xavier-uniform matrices inside the skip
transformers, cross_attention.py:36-39; U(-1/sqrt(fan_in), +) elsewhere; U[0,1)
learned PEs, position_encoding.py:150-151) except that LayerNorm affine
parameters and attention biases are perturbed away from (1, 0) so a kernel that
drops a bias or a gamma cannot pass parity by accident.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

NFEATS = 263          # configs/base.yaml DATASET.HUMANML3D (HumanML3D feature width)
NJOINTS = 22
TEXT_DIM = 768        # configs/modules/denoiser.yaml:4
MAX_PE = 500          # position_encoding.py:140


@dataclass
class ModelDims:
    """Shape hyper-parameters of config_mld_humanml3d.yaml (model.* block)."""
    latent_dim: int = 256
    latent_size: int = 1
    ff_size: int = 1024
    num_layers: int = 9
    num_heads: int = 4
    nfeats: int = NFEATS
    text_dim: int = TEXT_DIM

    @property
    def num_block(self) -> int:
        return (self.num_layers - 1) // 2


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _uniform(seed, name, shape, bound):
    return _rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32)


def _xavier(seed, name, shape):
    fan_out, fan_in = shape
    return _uniform(seed, name, shape, float(np.sqrt(6.0 / (fan_in + fan_out))))


def _linear(sd, seed, prefix, n_out, n_in, xavier=False):
    w = f"{prefix}.weight"
    sd[w] = _xavier(seed, w, (n_out, n_in)) if xavier else _uniform(seed, w, (n_out, n_in), 1.0 / np.sqrt(n_in))
    sd[f"{prefix}.bias"] = _uniform(seed, f"{prefix}.bias", (n_out,), 1.0 / np.sqrt(n_in))


def _norm(sd, seed, prefix, d):
    sd[f"{prefix}.weight"] = (1.0 + _uniform(seed, f"{prefix}.weight", (d,), 0.1)).astype(np.float32)
    sd[f"{prefix}.bias"] = _uniform(seed, f"{prefix}.bias", (d,), 0.05)


def _mha(sd, seed, prefix, d):
    sd[f"{prefix}.in_proj_weight"] = _xavier(seed, f"{prefix}.in_proj_weight", (3 * d, d))
    sd[f"{prefix}.in_proj_bias"] = _uniform(seed, f"{prefix}.in_proj_bias", (3 * d,), 0.02)
    sd[f"{prefix}.out_proj.weight"] = _xavier(seed, f"{prefix}.out_proj.weight", (d, d))
    sd[f"{prefix}.out_proj.bias"] = _uniform(seed, f"{prefix}.out_proj.bias", (d,), 0.02)


def block_names(num_block: int) -> List[str]:
    """Layer prefixes of a Skip transformer in execution order (cross_attention.py:41-60)."""
    return ([f"input_blocks.{i}" for i in range(num_block)] + ["middle_block"] +
            [f"output_blocks.{i}" for i in range(num_block)])


def _skip_transformer(sd, seed, prefix, dims: ModelDims, decoder: bool):
    d, ff = dims.latent_dim, dims.ff_size
    for blk in block_names(dims.num_block):
        p = f"{prefix}.{blk}"
        _mha(sd, seed, f"{p}.self_attn", d)
        if decoder:
            _mha(sd, seed, f"{p}.multihead_attn", d)
        _linear(sd, seed, f"{p}.linear1", ff, d, xavier=True)
        _linear(sd, seed, f"{p}.linear2", d, ff, xavier=True)
        for n in (("norm1", "norm2", "norm3") if decoder else ("norm1", "norm2")):
            _norm(sd, seed, f"{p}.{n}", d)
    for i in range(dims.num_block):
        _linear(sd, seed, f"{prefix}.linear_blocks.{i}", d, 2 * d, xavier=True)
    _norm(sd, seed, f"{prefix}.norm", d)


def make_denoiser_state_dict(seed: int = 0, dims: ModelDims = ModelDims(), condition: str = "text",
                             nclasses: int = 12) -> Dict[str, np.ndarray]:
    """Synthetic ``MldDenoiser`` weights (trans_enc + skip, learned PE).

    condition 'text'  : 768-wide timestep embedding + ReLU/Linear text projection (mld_denoiser.py:57-68)
    condition 'action': latent-wide timestep embedding + ``EmbedAction`` table (mld_denoiser.py:69-77,231-246)
    """
    sd: Dict[str, np.ndarray] = {}
    d = dims.latent_dim
    if condition == "action":
        _linear(sd, seed, "time_embedding.linear_1", d, d)
        _linear(sd, seed, "time_embedding.linear_2", d, d)
        sd["emb_proj.action_embedding"] = _xavier(seed, "emb_proj.action_embedding", (nclasses, d))
    else:
        _linear(sd, seed, "time_embedding.linear_1", d, dims.text_dim)
        _linear(sd, seed, "time_embedding.linear_2", d, d)
        _linear(sd, seed, "emb_proj.1", d, dims.text_dim)
    sd["query_pos.pe"] = _rng(seed, "query_pos.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    sd["mem_pos.pe"] = _rng(seed, "mem_pos.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    _skip_transformer(sd, seed, "encoder", dims, decoder=False)
    return sd


def trained_like(sd: Dict[str, np.ndarray], seed: int = 11) -> Dict[str, np.ndarray]:
    """A SECOND weight family with the statistics trained transformers show and xavier-uniform initialisation does not (tolerances must
    not be tuned to one distribution): LayerNorm gains ~ N(1, 0.3) (clipped to [0.15, 2.5]) and biases ~ N(0, 0.1); every weight matrix gets heavy-tailed ROWS
    (a log-normal factor per output row, sigma 0.6, and one row in 64 another x 4); biases ~ N(0, 0.05) on top of what they were; the denoiser's final norm
    (`encoder.norm.*`) shrinks its gain to ~0.3 so the predicted noise is several times smaller than with the first family.  (The sampled LATENTS are not: the reference
    modules on this family end at |latent| max 74.1 against ~80 -- a random-weight denoiser leaves DDIM's x0 estimate at ~ 14 x_T either way; the small- and large-latent
    regimes have their own fixtures, `latent_scale_case`.)
    Same keys and shapes as its input; deterministic in (seed, key)."""
    out: Dict[str, np.ndarray] = {}
    for k, v in sd.items():
        g = _rng(seed, "tl:" + k)
        a = np.array(v, np.float32, copy=True)
        is_norm = (".norm" in k or k.startswith("norm") or "norm." in k) and a.ndim == 1
        if is_norm and k.endswith("weight"):
            a = np.clip(g.normal(1.0, 0.3, a.shape), 0.15, 2.5).astype(np.float32)
            if k.endswith("encoder.norm.weight"):
                a *= 0.3
        elif is_norm and k.endswith("bias"):
            a = g.normal(0.0, 0.1, a.shape).astype(np.float32)
        elif a.ndim == 2 and k.endswith("weight") and min(a.shape) >= 16:
            row = np.exp(g.normal(0.0, 0.6, (a.shape[0], 1))).astype(np.float32)
            row[g.random((a.shape[0], 1)) < 1.0 / 64.0] *= 4.0
            a = a * row / np.float32(np.exp(0.18))          # keep the mean square of the matrix where it was (E exp(2 N(0, 0.6)) = e^0.72, x the outlier rows)
        elif a.ndim == 1 and k.endswith("bias"):
            a = a + g.normal(0.0, 0.05, a.shape).astype(np.float32)
        out[k] = np.ascontiguousarray(a, np.float32)
    return out


LATENT_SCALE_CASES = {"small": dict(final_gain=0.06, sigma=0.1), "large": dict(final_gain=1.0, sigma=6.5)}
LATENT_SCALE_LENGTHS = [64, 40, 57, 64, 23, 64, 11, 48]


def latent_scale_case(name: str):
    """(denoiser state dict, VAE state dict, batch) of the small- / large-latent fixtures (tests/golden/pipeline_b8_latent_scales.npz, generated from the reference's
    modules by oracle/make_golden_latent_scales.py).  Every other fixture ends at |latent| ~ 74-80: a random-weight denoiser does not predict the noise it is given, so
    DDIM's x0 estimate is ~ x_T / sqrt(alpha_bar_T) = 14 x_T whatever the weights.  The other regimes are reached through what a caller controls -- the start noise
    (mld.py:303, injected; scaled here) and the denoiser's final LayerNorm: "small" = second weight family with encoder.norm x 0.06 and start noise x 0.1 (|latent| max 7.5,
    rms 1.4: the decoder's frame-to-frame signal is no longer drowned by the per-sample cross-attention vector), "large" = second family, start noise x 6.5 (|latent| max 312)."""
    c = LATENT_SCALE_CASES[name]
    sdd, sdv = trained_like(make_denoiser_state_dict()), trained_like(make_vae_state_dict(), seed=12)
    sdd["encoder.norm.weight"] = (sdd["encoder.norm.weight"] * np.float32(c["final_gain"])).astype(np.float32)
    sdd["encoder.norm.bias"] = (sdd["encoder.norm.bias"] * np.float32(c["final_gain"])).astype(np.float32)
    b = make_batch(8, LATENT_SCALE_LENGTHS, seed=4321, max_len=64)
    b.init_latents = (b.init_latents * np.float32(c["sigma"])).astype(np.float32)
    return sdd, sdv, b


def make_novae_denoiser_state_dict(seed: int = 4, dims: ModelDims = ModelDims(latent_dim=512)) -> Dict[str, np.ndarray]:
    """Synthetic ``MldDenoiser`` weights for the diffusion-only variant (VAE_TYPE 'no', arch trans_dec, text condition;
    mld_denoiser.py:50-53,57-68,114-131): pose_embd / pose_proj, 9 plain TransformerDecoderLayers + final norm."""
    sd: Dict[str, np.ndarray] = {}
    d, ff = dims.latent_dim, dims.ff_size
    _linear(sd, seed, "pose_embd", d, dims.nfeats)
    _linear(sd, seed, "pose_proj", dims.nfeats, d)
    _linear(sd, seed, "time_embedding.linear_1", d, dims.text_dim)
    _linear(sd, seed, "time_embedding.linear_2", d, d)
    _linear(sd, seed, "emb_proj.1", d, dims.text_dim)
    sd["query_pos.pe"] = _rng(seed, "query_pos.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    sd["mem_pos.pe"] = _rng(seed, "mem_pos.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    for i in range(dims.num_layers):
        p = f"decoder.layers.{i}"
        _mha(sd, seed, p + ".self_attn", d)
        _mha(sd, seed, p + ".multihead_attn", d)
        _linear(sd, seed, p + ".linear1", ff, d, xavier=True)
        _linear(sd, seed, p + ".linear2", d, ff, xavier=True)
        for n in ("norm1", "norm2", "norm3"):
            _norm(sd, seed, f"{p}.{n}", d)
    _norm(sd, seed, "decoder.norm", d)
    return sd


def make_vae_state_dict(seed: int = 1, dims: ModelDims = ModelDims()) -> Dict[str, np.ndarray]:
    """Synthetic ``MldVae`` weights (arch encoder_decoder, PE_TYPE mld, MLP_DIST false)."""
    sd: Dict[str, np.ndarray] = {}
    d = dims.latent_dim
    sd["global_motion_token"] = _rng(seed, "global_motion_token").standard_normal(
        (2 * dims.latent_size, d)).astype(np.float32)
    sd["query_pos_encoder.pe"] = _rng(seed, "query_pos_encoder.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    sd["query_pos_decoder.pe"] = _rng(seed, "query_pos_decoder.pe").uniform(0, 1, (MAX_PE, 1, d)).astype(np.float32)
    _skip_transformer(sd, seed, "encoder", dims, decoder=False)
    _skip_transformer(sd, seed, "decoder", dims, decoder=True)
    _linear(sd, seed, "skel_embedding", d, dims.nfeats)
    _linear(sd, seed, "final_layer", dims.nfeats, d)
    return sd


def sinusoidal_pe(d_model: int, max_len: int = 5000) -> np.ndarray:
    """PositionalEncoding buffer (mld/models/operator/position_encoding_layer.py:15-23), [max_len, 1, d]."""
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d_model)).astype(np.float32)
    pe = np.zeros((max_len, d_model), np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe[:, None, :]


def make_actor_vae_state_dict(seed: int = 2, dims: ModelDims = ModelDims(nfeats=150), num_layers: int = 6) -> Dict[str, np.ndarray]:
    """Synthetic ``ActorVae`` weights (mld/models/architectures/actor_vae.py:21-70,77-125,176-207): stock
    nn.TransformerEncoder/DecoderLayer stacks, sinusoidal PE buffers, mu/logvar tokens."""
    sd: Dict[str, np.ndarray] = {}
    d, ff = dims.latent_dim, dims.ff_size
    sd["encoder.mu_token"] = _rng(seed, "encoder.mu_token").standard_normal(d).astype(np.float32)
    sd["encoder.logvar_token"] = _rng(seed, "encoder.logvar_token").standard_normal(d).astype(np.float32)
    _linear(sd, seed, "encoder.skel_embedding", d, dims.nfeats)
    sd["encoder.sequence_pos_encoding.pe"] = sinusoidal_pe(d)
    for i in range(num_layers):
        p = f"encoder.seqTransEncoder.layers.{i}"
        _mha(sd, seed, p + ".self_attn", d)
        _linear(sd, seed, p + ".linear1", ff, d, xavier=True)
        _linear(sd, seed, p + ".linear2", d, ff, xavier=True)
        _norm(sd, seed, p + ".norm1", d)
        _norm(sd, seed, p + ".norm2", d)
    sd["decoder.sequence_pos_encoding.pe"] = sinusoidal_pe(d)
    for i in range(num_layers):
        p = f"decoder.seqTransDecoder.layers.{i}"
        _mha(sd, seed, p + ".self_attn", d)
        _mha(sd, seed, p + ".multihead_attn", d)
        _linear(sd, seed, p + ".linear1", ff, d, xavier=True)
        _linear(sd, seed, p + ".linear2", d, ff, xavier=True)
        for n in ("norm1", "norm2", "norm3"):
            _norm(sd, seed, f"{p}.{n}", d)
    _linear(sd, seed, "decoder.final_layer", dims.nfeats, d)
    return sd


def make_action_batch(batch: int, nframes: int = 60, nclasses: int = 12, seed: int = 1234, dims: ModelDims = ModelDims()):
    """HumanAct12-shaped sampler inputs: labels uniform in [0, nclasses), N(0,1) start noise, fixed length."""
    g = _rng(seed, f"abatch{batch}")
    actions = g.integers(0, nclasses, size=batch).astype(np.int32)
    lat = g.standard_normal((batch, dims.latent_size, dims.latent_dim)).astype(np.float32)
    return actions, lat, [nframes] * batch


def make_mean_std(nfeats: int = NFEATS) -> Tuple[np.ndarray, np.ndarray]:
    """Physically scaled stand-ins for HumanML3D ``Mean.npy`` / ``Std.npy`` (get_data.py:38-40).

    Made-up magnitudes: small std on the root
    yaw / XZ velocity channels so the integrated trajectory stays physical; with
    mean=0, std=1 the joint metric is chaotic and the 1e-3 tolerance meaningless.
    """
    mean = np.zeros(nfeats, np.float32)
    std = np.full(nfeats, 0.3, np.float32)
    std[0] = 0.05
    std[1:3] = 0.03
    std[3] = 0.1
    mean[3] = 0.9
    return mean, std


@dataclass
class SyntheticBatch:
    text_emb: np.ndarray        # [2B, 1, 768]; rows 0..B-1 = the shared "" (unconditional) embedding
    init_latents: np.ndarray    # [B, 1, 256]
    lengths: List[int] = field(default_factory=list)


def make_batch(batch: int, lengths=None, seed: int = 1234, max_len: int = 196,
               dims: ModelDims = ModelDims()) -> SyntheticBatch:
    """HumanML3D-shaped sampler inputs: CLIP-like embeddings, N(0,1) start noise.

    ``lengths='ragged'`` draws multiples of 4 in [40, max_len]; ``None`` = all ``max_len``.
    Real CLIP ViT-L/14 text features have norm ~ 10-20 over 768 dims, i.e. O(0.5)
    per element; we draw N(0, 0.5^2).
    """
    g = _rng(seed, f"batch{batch}")
    uncond = (0.5 * g.standard_normal((1, 1, dims.text_dim))).astype(np.float32)
    cond = (0.5 * g.standard_normal((batch, 1, dims.text_dim))).astype(np.float32)
    text = np.concatenate([np.repeat(uncond, batch, 0), cond], 0)
    lat = g.standard_normal((batch, dims.latent_size, dims.latent_dim)).astype(np.float32)
    if lengths is None:
        lens = [max_len] * batch
    elif isinstance(lengths, str) and lengths == "ragged":
        lens = [int(x) for x in 4 * g.integers(10, max_len // 4 + 1, size=batch)]
    else:
        lens = [int(x) for x in lengths]
        assert len(lens) == batch
    return SyntheticBatch(text, lat, lens)
