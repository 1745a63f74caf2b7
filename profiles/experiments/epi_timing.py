"""Stamps inside epilogue_store of ONE bf16 conv instantiation (build: variant.sh et conv_bf16.hip -DCONV_TIMING -DEPI_TIMING
-DCONV_TIMING_KCH=32 -DCONV_TIMING_NCH=64 -DCONV_TIMING_H=32; run: CVAE_LIB=ab/et.so python profiles/experiments/epi_timing.py).
Per sampled workgroup and wave: cycles between consecutive stamps of tile 0 and tile 1 (conv_epilogue.h, EPI_STAMP):
entry | barrier | nb0 patch | nb0 reads+stores | nb1 patch | nb1 reads+stores | barrier | sums | barrier | M2 | barrier | bnpart."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402
from critic_vae_amd.train import FusedTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
vae = VariationalAutoencoder(max_batch=B, seed=0, precision="bf16").to(dev)
tr = FusedTrainer(vae)
x = torch.rand(B, 3, 64, 64, device=dev)
pred = torch.rand(B, 1, device=dev)
eps = torch.randn(B, 32, device=dev)
for _ in range(5):
    tr.step(x, pred, eps)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (16 * 4 * 32))()
vae.handle.lib.cvae_epi_dbg_read(buf)
names = ["barrier", "patch0", "st0", "patch1", "st1", "barrier", "sums", "barrier", "M2", "barrier", "bnpart"]
for g in range(0, 16, 4):
    for w in range(4):
        t = list(buf[(g * 4 + w) * 32:(g * 4 + w) * 32 + 24])
        if t[0] == 0:
            continue
        for tl in range(2):
            s = t[tl * 12:tl * 12 + 12]
            print(f"wg {64 * g:4d} wave {w} tile {tl}: " + "  ".join(f"{n} {s[i + 1] - s[i]}" for i, n in enumerate(names)) + f"   total {s[11] - s[0]}")
        print(f"                    tile0 end -> tile1 entry {t[12] - t[11]}   whole epilogue {t[23] - t[0]}")
