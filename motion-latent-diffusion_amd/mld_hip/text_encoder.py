"""Text encoders at the INPUT boundary of the hot path.

``MldTextEncoder`` keeps the frozen CLIP ViT-L/14 text tower on PyTorch-ROCm exactly as the reference does
(mld/models/architectures/mld_clip.py:17-90) -- it is not part of the HIP engine (BASELINE.json:
"the frozen CLIP text encoder run once on PyTorch-ROCm").  It also absorbs the transformers>=5 API
change (``get_text_features`` returns a ModelOutput there, which breaks the reference's ``.unsqueeze(1)``).

``SyntheticTextEncoder`` exists because no CLIP weights are reachable offline: a deterministic stand-in
with CLIP-like statistics so ``MLD.forward({"text": ..., "length": ...})`` can be exercised end to end.
"""
from __future__ import annotations

import os
import zlib
from typing import List

import numpy as np
import torch
from torch import nn


class MldTextEncoder(nn.Module):
    def __init__(self, modelpath: str, finetune: bool = False, last_hidden_state: bool = False,
                 latent_dim: list = [1, 256]) -> None:
        super().__init__()
        if last_hidden_state:
            raise NotImplementedError("last_hidden_state=True (token-level CLIP states) is not used by the MLD configs")
        if not os.path.isdir(modelpath):
            raise FileNotFoundError(f"CLIP weights not found at {modelpath!r} (configs/assets.yaml model.clip_path). "
                                    "Offline runs can pass text_encoder=SyntheticTextEncoder() to mld_hip.MLD.")
        from transformers import AutoModel, AutoTokenizer
        self.latent_dim = latent_dim
        self.tokenizer = AutoTokenizer.from_pretrained(modelpath)
        self.text_model = AutoModel.from_pretrained(modelpath)
        if not finetune:
            self.text_model.eval()
            for p in self.text_model.parameters():
                p.requires_grad = False
        self.max_length = self.tokenizer.model_max_length
        self.text_encoded_dim = self.text_model.config.text_config.hidden_size
        self.name = "clip"

    @torch.no_grad()
    def forward(self, texts: List[str]):
        ids = self.tokenizer(texts, padding="max_length", truncation=True, max_length=self.max_length,
                             return_tensors="pt").input_ids[:, : self.tokenizer.model_max_length]
        out = self.text_model.get_text_features(ids.to(next(self.text_model.parameters()).device))
        if not torch.is_tensor(out):                       # transformers >= 5: BaseModelOutputWithPooling
            out = out.pooler_output if getattr(out, "pooler_output", None) is not None else out[0]
        return out.unsqueeze(1).float()                    # [B, 1, 768]


class SyntheticTextEncoder(nn.Module):
    """Deterministic text -> [B, 1, 768] embedding (hash-seeded N(0, 0.5^2)); "" maps to one fixed vector."""

    def __init__(self, text_encoded_dim: int = 768, seed: int = 1234):
        super().__init__()
        self.text_encoded_dim = text_encoded_dim
        self.seed = seed
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)   # follows .to(device)
        self.name = "synthetic"

    def forward(self, texts: List[str]):
        rows = []
        for t in texts:
            g = np.random.Generator(np.random.PCG64([self.seed, zlib.crc32(t.encode())]))
            rows.append((0.5 * g.standard_normal(self.text_encoded_dim)).astype(np.float32))
        return torch.from_numpy(np.stack(rows)[:, None, :]).to(self._anchor.device)
