"""128-wide MS-SSIM level: band-streaming kernel (default) against the 16 x 64 tile kernel (CVAE_MS_STREAM=0) on the same frames.
The switch is read once per process, so run twice and compare the dumps:
    CVAE_MS_STREAM=1 python profiles/experiments/ms_stream_check.py /tmp/ms1.npz && CVAE_MS_STREAM=0 python profiles/experiments/ms_stream_check.py /tmp/ms0.npz
    python profiles/experiments/ms_stream_check.py /tmp/ms1.npz /tmp/ms0.npz
Expected: d_recon (= the F fields of all five levels through the backward pass) bit-identical — the same fma chains over the same products
per output — and the loss scalars equal up to the order of the per-plane partial sums (1e-7)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: max |diff| {d.max():.3e}  (max |value| {np.abs(a[k]).max():.3e})  bit-identical: {np.array_equal(a[k], b[k])}")
    sys.exit(0)

from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for B in (5, 64):
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, 128, 128, generator=g).to(dev)
    # a smooth-ish second image so that every level stays positive
    recon = (0.7 * x + 0.3 * torch.rand(B, 3, 128, 128, generator=g).to(dev)).contiguous()
    mu, logvar = torch.randn(B, 32, generator=g).to(dev) * 0.1, torch.randn(B, 32, generator=g).to(dev) * 0.1
    vae = VariationalAutoencoder(width=128, max_batch=B, seed=0).to(dev)
    h = vae.handle
    ws = torch.empty(h.workspace_bytes(B) // 4, device=dev)
    scal = torch.zeros(16, device=dev)
    d_recon, d_mu, d_lv = torch.empty_like(recon), torch.empty_like(mu), torch.empty_like(logvar)
    h.loss(B, x, mu, logvar, recon, ws, scal, d_recon, d_mu, d_lv)
    torch.cuda.synchronize()
    out[f"scalars_b{B}"] = scal.cpu().numpy()
    out[f"d_recon_b{B}"] = d_recon.cpu().numpy()
np.savez(sys.argv[1], **out)
print({k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
