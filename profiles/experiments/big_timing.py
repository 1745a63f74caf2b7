"""Phase timing of one instantiation of the 4 x 4 wave-tile bf16 conv kernel (build: variant.sh bt conv_bf16_big.hip -DBIG_TIMING -DBIG_T_KCH=256
-DBIG_T_NCH=128 -DBIG_T_H=8; run: CVAE_LIB=ab/bt.so python profiles/experiments/big_timing.py [B]): cycles per WORKGROUP, wave 0:
prologue | per item: stages (MFMA stream with everything interleaved) | chunk-closing barriers + first fragments | epilogue."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402
from critic_vae_amd.train import FusedTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
vae = VariationalAutoencoder(max_batch=B, seed=0, precision="bf16", width=W).to(dev) if W != 64 else VariationalAutoencoder(max_batch=B, seed=0, precision="bf16").to(dev)
tr = FusedTrainer(vae)
x, pred, eps = torch.rand(B, 3, W, W, device=dev), torch.rand(B, 1, device=dev), torch.randn(B, 32, device=dev)
for _ in range(5):
    tr.step(x, pred, eps)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 192)()
vae.handle.lib.cvae_big_dbg_read(buf)
rows = [list(buf[g * 12:g * 12 + 12]) for g in range(16) if buf[g * 12 + 5]]
if rows:
    print(f"first entry -> last exit {(max(r[9] for r in rows) - min(r[8] for r in rows)) / 100.0:.1f} us")
for g, t in enumerate(rows):
    n, st = max(t[0], 1), max(t[7], 1)
    print(f"wg {16 * g:4d}: {t[0]} items; prologue {t[1]}; per item: stages {t[2] // n} ({t[2] // n // st}/stage, ideal {st * 2560})  chunk close {t[3] // n}  epilogue {t[4] // n}  "
          f"| total {t[5]}  clock {t[5] / max(t[6], 1) * 0.1:.2f} GHz ({t[6] / 100.0:.1f} us; start +{(t[8] - min(r[8] for r in rows)) / 100.0:.1f} us)")
