"""mld_hip -- MI355X-native sampling engine for Motion Latent Diffusion.

Host-side mirror of the reference's plugin surface (instantiate_from_config targets) over the C ABI of
libmldhip.so.  Importing this package never touches the GPU; creating an engine without an MI355X raises.
See DESIGN.md / INTEGRATION.md at the repository root.
"""
from . import synthetic  # noqa: F401  (pure numpy; safe to import anywhere)

__all__ = ["synthetic", "config", "engine", "denoiser", "vae", "scheduler", "mld", "text_encoder", "datamodule", "dp", "checkpoint", "demo"]
