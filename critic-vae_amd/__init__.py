"""critic-vae_amd — MI355X-native Critic-VAE training step (hot path only).

Host side mirrors the reference's Python API for the path (vae_nets.py / vae.py:33-66); all
arithmetic runs in hand-written HIP kernels behind the C-ABI declared in include/cvae.h.
"""
from . import params  # noqa: F401
