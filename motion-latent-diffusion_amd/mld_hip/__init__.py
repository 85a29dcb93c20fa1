"""mld_hip -- MI355X-native sampling engine for Motion Latent Diffusion (host-side mirror of the
reference's plugin surface over the C ABI of libmldhip.so).  See DESIGN.md / INTEGRATION.md."""
from . import synthetic  # noqa: F401  (pure numpy; safe to import anywhere)

__all__ = ["synthetic"]
