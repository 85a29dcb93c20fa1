"""ctypes binding of include/mldhip.h -- the only door between Python and the HIP engine.

There is no CPU implementation behind this module: if ``libmldhip.so`` is missing, was not built
for gfx950, or no MI355X is visible, loading / ``Engine()`` raises.  (The test-suite's functional
simulator is injected explicitly by tests via ``load_library(path)``; nothing here looks for it.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libmldhip.so")
ABI_VERSION = 5


class MldHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mldhip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    """Mirror of ``mldhip_config`` (include/mldhip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("latent_dim", C.c_int32), ("latent_size", C.c_int32), ("ff_size", C.c_int32),
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("nfeats", C.c_int32), ("njoints", C.c_int32),
        ("text_dim", C.c_int32), ("max_batch", C.c_int32), ("max_frames", C.c_int32),
        ("num_train_timesteps", C.c_int32), ("num_inference_steps", C.c_int32), ("steps_offset", C.c_int32),
        ("set_alpha_to_one", C.c_int32), ("beta_start", C.c_float), ("beta_end", C.c_float),
        ("guidance_scale", C.c_float), ("precision", C.c_int32), ("use_graph", C.c_int32),
        ("condition", C.c_int32), ("nclasses", C.c_int32), ("vae_arch", C.c_int32), ("vae_num_layers", C.c_int32),
        ("denoiser_arch", C.c_int32), ("scheduler_type", C.c_int32), ("max_in_flight", C.c_int32),
    ]


class Request(C.Structure):
    """Mirror of ``mldhip_request`` (include/mldhip.h): one request of ``mldhip_sample_many``."""
    _fields_ = [("text_emb_dev", C.c_void_p), ("actions_host", C.POINTER(C.c_int32)), ("init_latents_dev", C.c_void_p),
                ("lengths_host", C.POINTER(C.c_int32)), ("B", C.c_int32), ("latents_out_dev", C.c_void_p),
                ("feats_out_dev", C.c_void_p), ("joints_out_dev", C.c_void_p)]


class NumericInfo(C.Structure):
    """Mirror of ``mldhip_numeric_info`` (include/mldhip.h, "Range contract" of the split-f16 mode)."""
    _fields_ = [("struct_size", C.c_int32), ("probed", C.c_int32), ("loop_split_ok", C.c_int32), ("decode_split_ok", C.c_int32),
                ("probe_err_loop", C.c_float), ("probe_err_decode", C.c_float), ("nonfinite_values", C.c_int64),
                ("decode_half_ok", C.c_int32), ("probe_err_decode_half", C.c_float), ("cluster_loop", C.c_int32), ("reserved", C.c_int32)]


PROBE_TOL = 6e-6                         # MLDHIP_PROBE_TOL
PROBE_TOL_HALF = 3e-5                     # MLDHIP_PROBE_TOL_HALF
COND_TEXT, COND_ACTION = 0, 1            # MLDHIP_COND_*
VAE_MLD, VAE_ACTOR, VAE_NONE = 0, 1, 2   # MLDHIP_VAE_*
ARCH_TRANS_ENC, ARCH_TRANS_DEC = 0, 1    # MLDHIP_ARCH_*
SCHED_DDIM, SCHED_DDPM = 0, 1            # MLDHIP_SCHED_*


_SYMBOLS = {
    # name: (restype, argtypes)
    "mldhip_abi_version": (C.c_int, []),
    "mldhip_default_config": (None, [C.POINTER(Config)]),
    "mldhip_create": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p)]),
    "mldhip_destroy": (None, [C.c_void_p]),
    "mldhip_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32]),
    "mldhip_finalize_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mldhip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mldhip_missing_keys": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mldhip_numeric_status": (C.c_int, [C.c_void_p, C.POINTER(NumericInfo)]),
    "mldhip_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "mldhip_sample_many": (C.c_int, [C.c_void_p, C.POINTER(Request), C.c_int32, C.c_void_p]),
    "mldhip_denoiser_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mldhip_sample_action": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mldhip_denoiser_forward_action": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_void_p,
                                                 C.c_void_p]),
    "mldhip_sample_novae": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "mldhip_denoiser_forward_novae": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_int32,
                                                C.c_int32, C.c_void_p, C.c_void_p]),
    "mldhip_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "mldhip_philox_normal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p]),
    "mldhip_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p]),
    "mldhip_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mldhip_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mldhip_feats2joints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mldhip_get_timesteps": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    "mldhip_get_alphas_cumprod": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
    "mldhip_get_launch_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mldhip_last_error": (C.c_char_p, [C.c_void_p]),
}


# measurement hooks: include/mldhip_hooks.h, exported by libmldhip_hooks.so only (`make -C csrc hooks`); bound when the loaded library has them
_HOOK_SYMBOLS = {
    "mldhip_profile_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    "mldhip_profile_trace": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.c_int64, C.c_void_p]),
}
HOOKS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmldhip_hooks.so")


def hooks_library() -> C.CDLL:
    """The hooks build of the engine (tools/ only): the production surface + mldhip_profile_kernel / mldhip_profile_trace / option "fused_dbg"."""
    if not os.path.exists(HOOKS_LIB):
        raise FileNotFoundError(f"{HOOKS_LIB} not found: build it with `make -C motion-latent-diffusion_amd/csrc hooks`")
    return load_library(HOOKS_LIB)


def exported_symbols() -> List[str]:
    return sorted(_SYMBOLS)


def load_library(path: Optional[str] = None) -> C.CDLL:
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "mld_hip has no CPU fallback.")
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and finds it by file name; if the
    # system copy were loaded first the process would end up with two HIP runtimes.  Import torch first
    # so libmldhip binds to the one runtime torch uses (device memory and streams are shared with it).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in _SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _HOOK_SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.mldhip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI version mismatch: library {lib.mldhip_abi_version()}, binding {ABI_VERSION}")
    return lib


def _ptr(x) -> int:
    """Device/host address of a torch tensor, numpy array, or raw int."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()          # torch.Tensor


class Engine:
    """Thin RAII wrapper over an ``mldhip_handle``."""

    def __init__(self, lib: Optional[C.CDLL] = None, device: int = 0, **overrides):
        self.lib = lib or load_library()
        self.cfg = Config()
        self.lib.mldhip_default_config(C.byref(self.cfg))
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise TypeError(f"unknown mldhip_config field {k!r}")
            setattr(self.cfg, k, v)
        self._h = C.c_void_p()
        rc = self.lib.mldhip_create(C.byref(self.cfg), device, C.byref(self._h))
        if rc != 0:
            raise MldHipError(rc, (self.lib.mldhip_last_error(None) or b"").decode())
        self.device = device

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int) -> int:
        if rc < 0:
            raise MldHipError(rc, (self.lib.mldhip_last_error(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.mldhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_tensor(self, key: str, array, on_device: bool = False) -> bool:
        """Returns False when the engine ignores the key (not on the sampling path)."""
        if isinstance(array, np.ndarray):
            array = np.ascontiguousarray(array, dtype=np.float32)
            shape = array.shape
        else:
            array = array.detach().contiguous().float()
            shape = tuple(array.shape)
            on_device = array.is_cuda
            if on_device:
                # mldhip_load_tensor copies on the null stream; torch side streams are non-blocking, so a parameter that was
                # just written on the caller's current stream (.to(device), load_state_dict) must have landed first
                import torch
                torch.cuda.current_stream(array.device).synchronize()
        shp = (C.c_int64 * len(shape))(*shape)
        rc = self._check(self.lib.mldhip_load_tensor(self._h, key.encode(), _ptr(array), shp, len(shape), 0,
                                                     1 if on_device else 0))
        return rc == 0

    def load_state_dict(self, tensors: Dict[str, object], prefix: str = "") -> List[str]:
        ignored = []
        for k, v in tensors.items():
            if not self.load_tensor(prefix + k, v):
                ignored.append(prefix + k)
        return ignored

    def missing_keys(self) -> List[str]:
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.mldhip_missing_keys(self._h, buf, len(buf))
        if n <= 0:
            return []
        return [s.decode() for s in buf.raw.split(b"\0") if s][:n]

    def set_option(self, name: str, value: int):
        """Per-handle tuning option (include/mldhip.h: loop_kernel, strip_min_rows, gemm_small_m)."""
        self._check(self.lib.mldhip_set_option(self._h, name.encode(), int(value)))
        if name == "range_probe":      # the probe is part of finalize: the C side un-finalizes the handle (advisor r4: the next sample failed with ESTATE)
            self._dirty = True

    def finalize(self, stream: int = 0):
        self._check(self.lib.mldhip_finalize_weights(self._h, stream))

    # ------------------------------------------------------------------ ops (raw pointers in, nothing allocated here)
    def sample(self, text_emb, init_latents, lengths: Sequence[int], latents_out=None, feats_out=None, joints_out=None,
               stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        self._check(self.lib.mldhip_sample(self._h, _ptr(text_emb), _ptr(init_latents), lens, len(lengths),
                                           _ptr(latents_out), _ptr(feats_out), _ptr(joints_out), stream))

    def sample_many(self, requests: Sequence[dict], stream: int = 0):
        """Several requests as ONE chain (mldhip_sample_many).  Each request is a dict with ``lengths`` and ``init_latents``,
        ``text_emb`` (text engines) or ``actions`` (action engines), and optional ``latents_out`` / ``feats_out`` / ``joints_out``."""
        arr = (Request * len(requests))()
        keep = []
        for r, q in zip(arr, requests):
            lens = (C.c_int32 * len(q["lengths"]))(*[int(x) for x in q["lengths"]])
            keep.append(lens)
            r.lengths_host, r.B = lens, len(q["lengths"])
            r.text_emb_dev = _ptr(q.get("text_emb")) or None
            if q.get("actions") is not None:
                acts = (C.c_int32 * len(q["actions"]))(*[int(x) for x in q["actions"]])
                keep.append(acts)
                r.actions_host = acts
            r.init_latents_dev = _ptr(q["init_latents"]) or None
            r.latents_out_dev = _ptr(q.get("latents_out")) or None
            r.feats_out_dev = _ptr(q.get("feats_out")) or None
            r.joints_out_dev = _ptr(q.get("joints_out")) or None
        self._check(self.lib.mldhip_sample_many(self._h, arr, len(requests), stream))

    def denoiser_forward(self, sample, timestep: int, text_emb, R: int, out, stream: int = 0):
        self._check(self.lib.mldhip_denoiser_forward(self._h, _ptr(sample), int(timestep), _ptr(text_emb), R, _ptr(out), stream))

    def sample_action(self, actions: Sequence[int], init_latents, lengths: Sequence[int], latents_out=None, feats_out=None,
                      stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        acts = (C.c_int32 * len(actions))(*[int(x) for x in actions])
        if len(actions) != len(lengths):
            raise ValueError("actions and lengths must have one entry per motion")
        self._check(self.lib.mldhip_sample_action(self._h, acts, _ptr(init_latents), lens, len(lengths), _ptr(latents_out),
                                                  _ptr(feats_out), stream))

    def denoiser_forward_action(self, sample, timestep: int, actions: Sequence[int], out, stream: int = 0):
        acts = (C.c_int32 * len(actions))(*[int(x) for x in actions])
        self._check(self.lib.mldhip_denoiser_forward_action(self._h, _ptr(sample), int(timestep), acts, len(actions), _ptr(out), stream))

    def sample_novae(self, text_emb, init_latents, lengths: Sequence[int], step_noise=None, seed: int = 0, feats_out=None,
                     joints_out=None, stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        self._check(self.lib.mldhip_sample_novae(self._h, _ptr(text_emb), _ptr(init_latents), lens, len(lengths), _ptr(step_noise),
                                                 int(seed), _ptr(feats_out), _ptr(joints_out), stream))

    def denoiser_forward_novae(self, sample, timestep: int, text_emb, lengths: Sequence[int], T: int, out, stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        self._check(self.lib.mldhip_denoiser_forward_novae(self._h, _ptr(sample), int(timestep), _ptr(text_emb), lens, len(lengths),
                                                           int(T), _ptr(out), stream))

    def ddpm_step(self, eps, timestep: int, sample, noise, prev_sample, n: int, seed: int = 0, step_index: int = 0, stream: int = 0):
        self._check(self.lib.mldhip_ddpm_step(self._h, _ptr(eps), int(timestep), _ptr(sample), _ptr(noise), int(seed), int(step_index),
                                              _ptr(prev_sample), n, stream))

    def philox_normal(self, out, n: int, seed: int, step_index: int, stream: int = 0):
        self._check(self.lib.mldhip_philox_normal(self._h, _ptr(out), n, int(seed), int(step_index), stream))

    def vae_decode(self, z, lengths: Sequence[int], feats_out, stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        self._check(self.lib.mldhip_vae_decode(self._h, _ptr(z), lens, len(lengths), _ptr(feats_out), stream))

    def vae_encode(self, feats, lengths: Sequence[int], T: int, eps, latent_out, mu_out, logvar_out, stream: int = 0):
        lens = (C.c_int32 * len(lengths))(*[int(x) for x in lengths])
        self._check(self.lib.mldhip_vae_encode(self._h, _ptr(feats), lens, len(lengths), int(T), _ptr(eps), _ptr(latent_out),
                                               _ptr(mu_out), _ptr(logvar_out), stream))

    def ddim_step(self, eps, timestep: int, sample, prev_sample, n: int, stream: int = 0):
        self._check(self.lib.mldhip_ddim_step(self._h, _ptr(eps), int(timestep), _ptr(sample), _ptr(prev_sample), n, stream))

    def feats2joints(self, feats, B: int, T: int, joints_out, stream: int = 0):
        self._check(self.lib.mldhip_feats2joints(self._h, _ptr(feats), B, T, _ptr(joints_out), stream))

    def _hook(self, name):
        fn = getattr(self.lib, name, None)
        if fn is None:
            raise MldHipError(-3, f"{name} is a measurement hook: load the hooks build (mld_hip._lib.hooks_library(), `make -C csrc hooks`)")
        return fn

    def profile_kernel(self, name: str, B: int, T: int, iters: int, stream: int = 0) -> float:
        """Enqueue one kernel `iters` times; returns its algorithmic FLOPs per launch."""
        fl = C.c_double(0.0)
        self._check(self._hook("mldhip_profile_kernel")(self._h, name.encode(), B, T, iters, C.byref(fl), stream))
        return fl.value

    def profile_trace(self, name: str, B: int, T: int, stream: int = 0) -> np.ndarray:
        """[workgroups, 8 waves, 8] uint64 timestamps of one traced den_* launch (measurement only)."""
        cap = 512 * 64
        buf = (C.c_uint64 * cap)()
        n = self._check(self._hook("mldhip_profile_trace")(self._h, name.encode(), B, T, buf, cap, stream))
        return np.ctypeslib.as_array(buf).reshape(-1, 8, 8)[:n].copy()

    def timesteps(self) -> np.ndarray:
        n = self.cfg.num_inference_steps
        buf = (C.c_int32 * n)()
        self._check(self.lib.mldhip_get_timesteps(self._h, buf, n))
        return np.array(buf[:], dtype=np.int64)

    def alphas_cumprod(self) -> np.ndarray:
        n = self.cfg.num_train_timesteps
        buf = (C.c_float * n)()
        self._check(self.lib.mldhip_get_alphas_cumprod(self._h, buf, n))
        return np.array(buf[:], dtype=np.float32)

    def numeric_status(self) -> dict:
        """Range contract of the split-f16 mode (mldhip_numeric_status): what finalize's probe decided and how many non-finite
        latents / joints values the sample calls since the previous query produced.  Synchronises the device."""
        info = NumericInfo()
        info.struct_size = C.sizeof(NumericInfo)
        self._check(self.lib.mldhip_numeric_status(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in NumericInfo._fields_ if k != "struct_size"}

    def launch_counts(self) -> List[int]:
        buf = (C.c_int32 * 3)()
        self._check(self.lib.mldhip_get_launch_counts(self._h, buf))
        return list(buf[:])
