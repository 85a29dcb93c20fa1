"""sys.path helper: the product package lives in a directory whose name is not an identifier."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(REPO, "motion-latent-diffusion_amd")
for p in (REPO, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)
