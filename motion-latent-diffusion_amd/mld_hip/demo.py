"""CLI mirroring the text-to-motion branch of the reference's demo.py (:40-50,:166-194):

    python -m mld_hip.demo --cfg configs/config_mld_humanml3d.yaml --example demo/example.txt --out_dir results

Reads "<length> <prompt>" lines (mld/utils/demo_utils.py:6-20), samples on the MI355X engine and writes
``Example_<len>_batch0_<i>.npy`` files of shape (nframe, 22, 3) plus the prompt as .txt.  Offline it
falls back to synthetic weights / the synthetic text encoder and says so (no checkpoints are reachable)."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch


def load_example_input(txt_path):
    texts, lens = [], []
    with open(txt_path, "r") as f:
        for line in f:
            s = line.strip()
            if not s:
                continue
            head = s.split(" ")[0]
            lens.append(int(head))
            texts.append(s[len(head) + 1:])
    return texts, lens


def main(argv=None):
    from .config import load_config
    from .datamodule import HipDataModule
    from .mld import MLD
    from .text_encoder import SyntheticTextEncoder

    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=None)
    ap.add_argument("--cfg_assets", default=None)
    ap.add_argument("--example", required=True)
    ap.add_argument("--out_dir", default="./results")
    ap.add_argument("--replication", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args(argv)
    cfg = load_config(a.cfg, a.cfg_assets)
    texts, lengths = load_example_input(a.example)
    dev = torch.device("cuda:0")
    dm = HipDataModule(cfg)
    try:
        model = MLD(cfg, dm)
        note = "CLIP text encoder"
    except FileNotFoundError:
        model = MLD(cfg, dm, text_encoder=SyntheticTextEncoder())
        note = "SYNTHETIC text encoder (no CLIP weights on disk)"
    ckpt = cfg.TEST.CHECKPOINTS
    if os.path.exists(ckpt):
        from .checkpoint import load_lightning_state_dict       # works without pytorch_lightning / omegaconf installed
        model.load_state_dict(load_lightning_state_dict(ckpt), strict=True)
    else:
        note += "; SYNTHETIC weights (checkpoint %s not found)" % ckpt
    model.to(dev).eval()
    torch.manual_seed(a.seed)
    os.makedirs(a.out_dir, exist_ok=True)
    print("mld_hip demo:", note, "| stats:", dm.stats)
    for rep in range(a.replication):
        joints = model({"length": lengths, "text": texts})
        for i, j in enumerate(joints):
            path = os.path.join(a.out_dir, f"Example_{lengths[i]}_batch{rep}_{i}.npy")
            np.save(path, j.numpy())
            with open(path.replace(".npy", ".txt"), "w") as f:
                f.write(texts[i])
            print("  wrote", path, tuple(j.shape))


if __name__ == "__main__":
    main()
