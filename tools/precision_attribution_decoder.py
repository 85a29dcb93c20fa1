"""Per-GEMM-class error attribution of cheaper operand formats for the DECODER: the motion-VAE decode
(mld_vae.py:186-248, cross_attention.py:66-125,323-345) runs ONCE behind the reverse loop, so its arithmetic error is not multiplied by
guidance x 50 steps -- which of its GEMM classes can leave split-f16 x3 (3 matrix instructions per product, Q|K|V kept as fp32) before the
<= 1e-3 joint contract breaks?

CPU-only, oracle-side EMULATION (numpy): the oracle's `vae_decode` with ONE class computed on cheaper operands (fp32 accumulation), everything
else exact fp32; then the candidate MIXES.  Classes are recognised by the weight tensor a product multiplies (address ranges of the state dict's
arrays) or, for the two attention products, by operand rank:

  in_proj   self_attn.in_proj_weight (Q | K | V of the frame rows)          out_proj   self_attn.out_proj
  linear1 / linear2 (feed-forward)      skip (linear_blocks, K = 512)         final (final_layer, N = 263)
  qk (Q K^T)     pv (softmax . V)       qkv_store = Q|K|V rounded to ONE half per element when they are stored (2 B instead of 4 B per element)
  (the 1-key cross-attention vector is a per-sample [1, 256] product: kept exact, as the engine computes it once per sample)

Formats:  f16 = both operands rounded to IEEE half, 1 matrix instruction;   a16w32 = half activations x split (hi + lo) weights, 2 instructions;
          a32w16 = split activations x half weights, 2 instructions;         x3 = today's split x split (lo x lo dropped), 3 instructions.

Reported per variant: max-abs error of features and of joints against the fp64 decode of the same latents, on BOTH synthetic weight families
(first family: latents of the committed reference fixture pipeline_b64, |z| up to 80; second: pipeline_b8_trainedlike; and both families on UNIT-NORMAL latents,
where the per-sample cross-attention vector no longer drowns the frame-to-frame signal the self-attention carries -- the harder case) and on the decoder-relevant range-contract
weight sets of tests/test_gpu_parity.py::_scaled_weights.

  python tools/precision_attribution_decoder.py [--batch 6] [--out profiles/r06_decoder_precision.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "motion-latent-diffusion_amd")]
import numpy as np
from mld_hip import synthetic as syn
from oracle import mld_oracle as O

F16MAX = 65504.0


def h(x):
    return np.clip(x, -F16MAX, F16MAX).astype(np.float16).astype(np.float32)


def split(x):
    hi = h(x)
    return hi, (x - hi).astype(np.float16).astype(np.float32)


def mm(fmt, a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    if fmt == "f32":
        return np.matmul(a, b)
    if fmt == "f16":
        return np.matmul(h(a), h(b))
    if fmt == "a16w32":                       # a = activations (rows), b = weights
        bh, bl = split(b); ah = h(a)
        return np.matmul(ah, bl) + np.matmul(ah, bh)
    if fmt == "a32w16":
        ah, al = split(a); bh = h(b)
        return np.matmul(al, bh) + np.matmul(ah, bh)
    if fmt == "x3":
        ah, al = split(a); bh, bl = split(b)
        return (np.matmul(al, bh) + np.matmul(ah, bl)) + np.matmul(ah, bh)
    raise ValueError(fmt)


def weight_class(key):
    if key.endswith("self_attn.in_proj_weight"): return "in_proj"
    if key.endswith("self_attn.out_proj.weight"): return "out_proj"
    if "multihead_attn" in key: return "cross"
    if key.endswith("linear1.weight"): return "linear1"
    if key.endswith("linear2.weight"): return "linear2"
    if "linear_blocks" in key: return "skip"
    if key.startswith("final_layer"): return "final"
    return None


class DecOps(O.NumpyOps):
    """fmt: {class: format}; classes not named run exact fp32.  'qkv_store': 'f16' rounds Q|K|V (bias included) to one half."""

    def __init__(self, sd, fmt):
        super().__init__(np.float32)
        self.fmt = dict(fmt)
        self.ranges = []
        for k, v in sd.items():
            c = weight_class(k)
            if c and v.ndim == 2:
                p = v.__array_interface__["data"][0]
                self.ranges.append((p, p + v.nbytes, c))

    def _cls(self, b):
        p = b.__array_interface__["data"][0]
        for lo, hi, c in self.ranges:
            if lo <= p < hi:
                if c == "in_proj":            # the oracle applies the packed in-projection as three row slices: q, k, v
                    return "in_proj_" + "qkv"[(p - lo) * 3 // (hi - lo)]
                return c
        return None

    def matmul(self, a, b):
        if np.ndim(b) == 4:                   # attention products: [N, H, L, hd] x [N, H, hd, S] (Q K^T) or [N, H, L, S] x [N, H, S, hd] (P V)
            c = "qk" if a.shape[-1] == 64 and b.shape[-2] == 64 and not getattr(self, "_next_is_pv", False) else "pv"
            self._next_is_pv = (c == "qk")
            if (c == "qk" and b.shape[-1] == 1) or (c == "pv" and b.shape[-2] == 1):
                return np.matmul(a, b)        # the 1-key cross-attention: exact (softmax = 1; the engine computes its vector once per sample)
            st = self.fmt.get("qkv_store")
            sq, sk, sv = (self.fmt.get("store_" + n, st) for n in "qkv")      # 'f16': the tensor is rounded to ONE half when stored
            if c == "qk":
                if sq == "f16": a = h(a)      # (the oracle scales q after the projection; the engine folds the scale in before rounding: same rounding class)
                if sk == "f16": b = h(b)
            elif sv == "f16":
                b = h(b)
            f = self.fmt.get(c, "f32")
            if c == "pv" and f in ("a16w32",): f = "f16"
            return mm(f, a, b)
        c = self._cls(b) if np.ndim(b) == 2 else None
        if c is None or c == "cross":
            return np.matmul(a, b)
        return mm(self.fmt.get(c, self.fmt.get(c[:7], "f32") if c.startswith("in_proj") else "f32"), a, b)


def scaled_vae(case):
    """decoder side of tests/test_gpu_parity.py::_scaled_weights"""
    sdv = syn.make_vae_state_dict()
    if case in ("ln_gain_up", "ln_gain_down"):
        f = np.float32(1024.0 if case == "ln_gain_up" else 2.0 ** -10)
        for k in ("decoder.input_blocks.1.norm2", "decoder.output_blocks.0.norm3"):
            sdv[k + ".weight"] = sdv[k + ".weight"] * f; sdv[k + ".bias"] = sdv[k + ".bias"] * f
    elif case == "ffn_hidden_huge":
        k = "decoder.input_blocks.2"
        sdv[k + ".linear1.weight"] = sdv[k + ".linear1.weight"] * np.float32(2.0 ** 14)
        sdv[k + ".linear2.weight"] = sdv[k + ".linear2.weight"] * np.float32(2.0 ** -14)
    elif case == "weights_tiny":
        sdv["decoder.input_blocks.1.self_attn.in_proj_weight"] = sdv["decoder.input_blocks.1.self_attn.in_proj_weight"] * np.float32(2.0 ** -12)
        sdv["decoder.linear_blocks.1.weight"] = sdv["decoder.linear_blocks.1.weight"] * np.float32(2.0 ** -12)
        sdv["decoder.linear_blocks.1.bias"] = sdv["decoder.linear_blocks.1.bias"] * np.float32(2.0 ** -12)
    return sdv


CLASSES = ["in_proj", "out_proj", "linear1", "linear2", "skip", "final", "qk", "pv"]
ALLX3 = {c: "x3" for c in CLASSES}


def mixes():
    m = {"today_all_x3": dict(ALLX3)}
    for f in ("f16", "a16w32", "a32w16"):
        for c in CLASSES:
            if c in ("qk", "pv") and f != "f16":
                continue
            d = dict(ALLX3); d[c] = f
            m[f"{c}:{f}"] = d
        m[f"all_gemms:{f}"] = {**ALLX3, **{c: f for c in CLASSES if c not in ("qk", "pv")}}
    d = dict(ALLX3); d["qkv_store"] = "f16"
    m["qkv_store:f16"] = d
    d = dict(ALLX3); d.update(qk="f16", pv="f16", qkv_store="f16")
    m["attention_all_f16(qkv_store+qk+pv)"] = d
    m["everything:f16"] = {**{c: "f16" for c in CLASSES}, "qkv_store": "f16"}
    m["everything:a16w32(+attn f16)"] = {**{c: "a16w32" for c in CLASSES if c not in ("qk", "pv")}, "qk": "f16", "pv": "f16", "qkv_store": "f16"}
    # candidate mixes (what a kernel could be built as)
    m["mixA: qkv_store f16 + attention f16, GEMMs x3"] = {**ALLX3, "qk": "f16", "pv": "f16", "qkv_store": "f16"}
    m["mixB: mixA + in_proj a16w32"] = {**ALLX3, "qk": "f16", "pv": "f16", "qkv_store": "f16", "in_proj": "a16w32"}
    m["mixC: mixB + linear1/linear2/out_proj a16w32"] = {**ALLX3, "qk": "f16", "pv": "f16", "qkv_store": "f16", "in_proj": "a16w32", "linear1": "a16w32",
                                                          "linear2": "a16w32", "out_proj": "a16w32"}
    m["mixD: mixC + skip/final a16w32"] = {**{c: "a16w32" for c in CLASSES if c not in ("qk", "pv")}, "qk": "f16", "pv": "f16", "qkv_store": "f16"}
    m["mixE: in_proj f16 + attention f16, rest x3"] = {**ALLX3, "qk": "f16", "pv": "f16", "qkv_store": "f16", "in_proj": "f16"}
    # P kept split in P V (a32w16 = split P x half V: 2 instructions); V stored split (exact) with Q | K as halves
    m["mixB': mixB with P split in P V"] = {**ALLX3, "qk": "f16", "pv": "a32w16", "qkv_store": "f16", "in_proj": "a16w32"}
    m["mixV: Q|K stored half, V stored split; in_proj q,k a16w32, v x3; Q K^T f16, P V x3"] = {**ALLX3, "qk": "f16", "pv": "x3", "store_q": "f16", "store_k": "f16",
                                                                                                "in_proj_q": "a16w32", "in_proj_k": "a16w32", "in_proj_v": "x3"}
    m["mixV2: mixV with the whole in_proj a16w32"] = {**ALLX3, "qk": "f16", "pv": "x3", "store_q": "f16", "store_k": "f16", "in_proj": "a16w32"}
    m["mixV3: mixV with P V as split P x half-hi V + half P x lo V (x3) but in_proj x3 everywhere"] = {**ALLX3, "qk": "f16", "pv": "x3", "store_q": "f16", "store_k": "f16"}
    m["mixF: GEMMs a16w32, attention x3 on fp32 Q|K|V"] = {**{c: "a16w32" for c in CLASSES if c not in ("qk", "pv")}, "qk": "x3", "pv": "x3"}
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_decoder_precision.json"))
    ap.add_argument("--only", default="", help="comma-separated substrings of variant names to run")
    ap.add_argument("--sets", default="", help="comma-separated substrings of set names to run (results are merged into an existing --out file)")
    a = ap.parse_args()
    mean, std = syn.make_mean_std()
    o64 = O.NumpyOps(np.float64)
    gold = os.path.join(ROOT, "tests", "golden")
    fam1 = np.load(os.path.join(gold, "pipeline_b64.npz"))["latents"][: a.batch]
    fam2 = np.load(os.path.join(gold, "pipeline_b8_trainedlike.npz"))["latents"][: a.batch]
    lens = ([196, 196, 120, 64, 196, 33, 196, 196] * 8)[: a.batch]
    sdv1 = syn.make_vae_state_dict()
    zu = syn._rng(31, "attr_unit_latents").standard_normal(fam1.shape).astype(np.float32)      # KL-regularised VAE latents are O(1): the cross-attention vector no longer drowns the frame-to-frame signal
    sets = {"family1": (sdv1, fam1), "family2_trained_like": (syn.trained_like(sdv1), fam2),
            "family1_unit_latents": (sdv1, zu), "family2_unit_latents": (syn.trained_like(sdv1), zu)}
    for case in ("ln_gain_up", "ln_gain_down", "weights_tiny"):      # (ffn_hidden_huge: the probe moves the decoder to fp32 -- not a split-format case)
        sets["range:" + case] = (scaled_vae(case), fam1)
    out = {"what": __doc__.split("\n\n")[0], "batch": a.batch, "lengths": lens, "tolerance_joints": 1e-3, "target_joints": 5e-4, "sets": {}}
    if a.sets:
        sets = {k: v for k, v in sets.items() if any(x in k for x in a.sets.split(","))}
        if os.path.exists(a.out):
            out["sets"] = json.load(open(a.out))["sets"]
    variants = mixes()
    if a.only:
        variants = {k: v for k, v in variants.items() if any(s in k for s in a.only.split(","))}
    for sname, (sdv, z) in sets.items():
        bv64 = O.to_backend(o64, sdv)
        f64 = np.asarray(O.vae_decode(o64, bv64, z.astype(np.float64), lens))
        j64 = np.asarray(O.feats2joints(o64, f64, mean.astype(np.float64), std.astype(np.float64)))
        valid = np.zeros(f64.shape[:2], bool)
        for i, n in enumerate(lens):
            valid[i, :n] = True
        tab = {}
        sd32 = {k: np.ascontiguousarray(v, np.float32) for k, v in sdv.items()}

        def run(fmt):
            ops = DecOps(sd32, fmt)
            f = np.asarray(O.vae_decode(ops, sd32, z.astype(np.float32), lens), np.float64)
            j = np.asarray(O.feats2joints(o64, f, mean.astype(np.float64), std.astype(np.float64)))
            return {"feats_max_abs": float(np.abs(f - f64)[valid].max()), "joints_max_abs": float(np.abs(j - j64)[valid].max())}
        t0 = time.time()
        tab["fp32"] = run({})
        print(sname, "fp32", tab["fp32"], f"{time.time() - t0:.0f}s", flush=True)
        for name, fmt in variants.items():
            t0 = time.time()
            r = run(fmt)
            r["meets_1e-3"] = r["joints_max_abs"] < 1e-3; r["meets_5e-4"] = r["joints_max_abs"] < 5e-4
            tab[name] = r
            print(sname, name, r, f"{time.time() - t0:.0f}s", flush=True)
        out["sets"][sname] = {"feats_absmax": float(np.abs(f64).max()), "joints_absmax": float(np.abs(j64).max()), "variants": tab}
        json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
