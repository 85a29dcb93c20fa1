"""``mld_hip.text_encoder.MldTextEncoder`` (the CLIP adapter at the input boundary, mld_clip.py:13-90) on a throw-away CLIP
directory: a random-init ``CLIPModel`` of the right widths plus a minimal ``CLIPTokenizer`` vocabulary written to tmp_path
(no released weights are reachable offline).  Checks the contract the hot path relies on: ``padding="max_length"`` to the
tokenizer's model_max_length (mld_clip.py:53-59), ``get_text_features`` = the PROJECTED pooled output (:75-76), ``unsqueeze(1)``
-> [B, 1, 768] (:78), identical rows for identical prompts (the "" half of a CFG batch, mld.py:224-231), frozen parameters
(:31-35), and the transformers >= 5 return type (a ModelOutput instead of a tensor)."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from mld_hip.text_encoder import MldTextEncoder  # noqa: E402


def make_clip_dir(root, hidden=768, layers=1):
    from transformers import CLIPConfig, CLIPModel, CLIPTokenizer
    d = os.path.join(str(root), "clip-vit-tiny-random")          # "clip" in the path, as the reference requires (mld_clip.py:39)
    os.makedirs(d, exist_ok=True)
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,!?'")
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = ["#version: 0.2", "w a", "wa l", "wal k</w>", "r u", "ru n</w>"]
    for tok in ("wa", "wal", "walk</w>", "ru", "run</w>"):
        vocab[tok] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("\n".join(merges) + "\n")
    CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"), model_max_length=77).save_pretrained(d)
    eos = vocab["<|endoftext|>"]
    cfg = CLIPConfig(text_config=dict(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=64, num_hidden_layers=layers,
                                      num_attention_heads=2, max_position_embeddings=77, projection_dim=768,
                                      bos_token_id=vocab["<|startoftext|>"], eos_token_id=eos, pad_token_id=eos),
                     vision_config=dict(hidden_size=32, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2, image_size=32,
                                        patch_size=16, projection_dim=768),
                     projection_dim=768)
    torch.manual_seed(0)
    CLIPModel(cfg).save_pretrained(d)
    return d, vocab


@pytest.fixture(scope="module")
def clip_dir(tmp_path_factory):
    return make_clip_dir(tmp_path_factory.mktemp("clip"))


def test_text_encoder_contract(clip_dir):
    d, vocab = clip_dir
    enc = MldTextEncoder(d, finetune=False, last_hidden_state=False, latent_dim=[1, 256])
    assert enc.name == "clip" and enc.max_length == 77 and enc.text_encoded_dim == 768
    assert not enc.text_model.training and all(not p.requires_grad for p in enc.text_model.parameters())     # frozen (:31-35)
    texts = ["", "a man walks.", "", "a person runs, then walks!"]
    out = enc(texts)
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32 and tuple(out.shape) == (4, 1, 768)    # [B, 1, 768] (:78)
    assert torch.equal(out[0], out[2]) and not torch.equal(out[0], out[1])                                    # "" rows identical
    # padding = max_length: every prompt becomes exactly 77 ids, start token first, EOS/pad afterwards (:53-59)
    ids = enc.tokenizer(texts, padding="max_length", truncation=True, max_length=enc.max_length, return_tensors="pt").input_ids
    assert tuple(ids.shape) == (4, 77) and int(ids[0, 0]) == vocab["<|startoftext|>"] and int(ids[0, 1]) == vocab["<|endoftext|>"]
    assert torch.all(ids[0, 1:] == vocab["<|endoftext|>"])
    # get_text_features = text_projection(pooled EOS state) -- whatever container this transformers version returns (:75-76)
    with torch.no_grad():
        pooled = enc.text_model.text_model(input_ids=ids).pooler_output
        want = enc.text_model.text_projection(pooled)
    assert torch.allclose(out[:, 0], want, atol=1e-5), float((out[:, 0] - want).abs().max())
    major = int(transformers.__version__.split(".")[0])
    raw = enc.text_model.get_text_features(ids)
    assert torch.is_tensor(raw) == (major < 5)           # >= 5 returns a ModelOutput: the adapter's pooler_output branch is exercised here
    # prompts longer than 77 tokens are truncated, not an error (:56)
    long = enc(["a " * 200])
    assert tuple(long.shape) == (1, 1, 768) and torch.isfinite(long).all()


def test_text_encoder_errors(tmp_path, clip_dir):
    with pytest.raises(FileNotFoundError):
        MldTextEncoder(str(tmp_path / "missing-clip"))
    with pytest.raises(NotImplementedError):
        MldTextEncoder(clip_dir[0], last_hidden_state=True)


def test_mld_forward_through_clip_adapter_on_the_simulator(clip_dir):
    """MLD.forward (mld.py:216-265) with the real adapter class in the text_encoder slot: the CFG batch is built as
    [""]*B + texts, encoded once, and fed to the engine (TEST-ONLY simulator here; the GPU twin is in test_gpu_parity.py)."""
    import simlib
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip import synthetic as syn
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD
    from oracle import mld_oracle as O

    eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=16, num_inference_steps=2, num_layers=simlib.SIM_LAYERS)
    key = E.inject_engine(eng, "inject:clip")
    cfg = C.load_config(overrides={"model.scheduler.num_inference_timesteps": 2, "model.denoiser.params.num_layers": simlib.SIM_LAYERS,
                                   "model.motion_vae.params.num_layers": simlib.SIM_LAYERS})
    enc = MldTextEncoder(clip_dir[0])
    model = MLD(cfg, HipDataModule(cfg, engine_key=key), text_encoder=enc, engine_key=key).eval()
    texts, lengths = ["a man walks.", "a person runs."], [12, 9]
    lat0 = torch.from_numpy(syn.make_batch(2, lengths).init_latents)
    joints = model({"text": texts, "length": lengths}, init_latents=lat0)
    assert [tuple(j.shape) for j in joints] == [(12, 22, 3), (9, 22, 3)]
    emb = enc([""] * 2 + texts).numpy()
    ops = O.NumpyOps(np.float32)
    mean, std = syn.make_mean_std()
    sdd, sdv = simlib.text_weights()
    jr = O.sample(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), emb,
                  lat0.numpy(), lengths, mean, std, steps=2)
    for i, n in enumerate(lengths):
        assert np.abs(joints[i].numpy() - jr[i, :n]).max() < 1e-3
    eng.close()


def test_checkpoint_load_reinjects_the_clip_weights(clip_dir):
    """MLD.load_state_dict as the demo uses it (demo.py:129-150 -> base.py:117-127): released checkpoints carry no usable CLIP tensors,
    so the model's OWN text-encoder tensors are put back under ``text_encoder.*`` before a STRICT load; evaluator nets (``t2m_*``) are
    dropped; whatever ``text_encoder.*`` a checkpoint does carry is ignored.  Run with the real adapter class (random-init CLIP)."""
    import simlib
    from mld_hip import config as C
    from mld_hip import engine as E
    from mld_hip.datamodule import HipDataModule
    from mld_hip.mld import MLD

    eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=2, max_frames=16, num_inference_steps=2, num_layers=simlib.SIM_LAYERS)
    key = E.inject_engine(eng, "inject:clip_ckpt")
    try:
        cfg = C.load_config(overrides={"model.scheduler.num_inference_timesteps": 2, "model.denoiser.params.num_layers": simlib.SIM_LAYERS,
                                       "model.motion_vae.params.num_layers": simlib.SIM_LAYERS})
        enc = MldTextEncoder(clip_dir[0])
        model = MLD(cfg, HipDataModule(cfg, engine_key=key), text_encoder=enc, engine_key=key).eval()
        clip_before = {k: v.clone() for k, v in enc.state_dict().items()}
        assert len(clip_before) > 10
        sdd, sdv = simlib.text_weights()
        ckpt = {**{"denoiser." + k: torch.from_numpy(v) for k, v in sdd.items()}, **{"vae." + k: torch.from_numpy(v) for k, v in sdv.items()}}
        bogus = next(iter(clip_before))
        ckpt["text_encoder." + bogus] = torch.full_like(clip_before[bogus], 7.0)         # stale CLIP tensor in the file: must not be loaded
        ckpt["t2m_moveencoder.main.0.weight"] = torch.zeros(3, 3)                          # evaluator net: must not make the strict load fail
        res = model.load_state_dict(ckpt, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        after = enc.state_dict()
        assert all(torch.equal(after[k], clip_before[k]) for k in clip_before)             # CLIP untouched, stale tensor ignored
        k0 = "encoder.norm.weight"
        assert torch.equal(model.denoiser.state_dict()[k0], torch.from_numpy(sdd[k0]))
        ckpt.pop("denoiser." + k0)
        with pytest.raises(RuntimeError):                                                  # strict: a missing network tensor is an error
            model.load_state_dict(ckpt, strict=True)
        joints = model({"text": ["a man walks."], "length": [9]})                          # and the loaded model samples
        assert tuple(joints[0].shape) == (9, 22, 3) and torch.isfinite(joints[0]).all()
    finally:
        E._engines.pop(key, None)
        eng.close()
