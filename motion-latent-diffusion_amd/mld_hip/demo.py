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


def write_motions(model, texts, lengths, out_dir, replication=1, task="Example", log=print):
    """Sample `replication` times and write the files the reference's demo writes (demo.py:166-194): per motion i
    ``<task>_<length_i>_batch<id>_<i>.npy`` = joints (nframe, 22, 3) float32 and the prompt under the same name with ``.txt``.
    The reference never advances ``id`` (demo.py:188), so its replications overwrite each other under ``batch0``; here replication
    r > 0 is kept as ``batch<r>`` and replication 0 carries the reference's exact names.  Returns the list of .npy paths."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for rep in range(replication):
        joints = model({"length": list(lengths), "text": list(texts)})
        for i, j in enumerate(joints):
            path = os.path.join(out_dir, f"{task}_{lengths[i]}_batch{rep}_{i}.npy")
            with open(path.replace(".npy", ".txt"), "w") as f:
                f.write(texts[i])
            np.save(path, j.detach().cpu().numpy())
            paths.append(path)
            log("  wrote", path, tuple(j.shape))
    return paths


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
    except (FileNotFoundError, OSError):
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
    print("mld_hip demo:", note, "| stats:", dm.stats)
    return write_motions(model, texts, lengths, a.out_dir, a.replication)


if __name__ == "__main__":
    main()
