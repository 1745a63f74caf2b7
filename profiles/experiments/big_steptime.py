"""Cycles per STEP (16 MFMAs = 512 ideal) of one chunk pair of the persistent big-tile bf16 conv kernel (build: variant.sh st conv_bf16_big.hip
-DBIG_TIMING -DBIG_STEPTIME -DBIG_T_KCH=256 -DBIG_T_NCH=128 -DBIG_T_H=8; run: CVAE_LIB=ab/st.so python profiles/experiments/big_steptime.py [B])."""
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
x, pred, eps = torch.rand(B, 3, 64, 64, device=dev), torch.rand(B, 1, device=dev), torch.randn(B, 32, device=dev)
for _ in range(5):
    tr.step(x, pred, eps)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
vae.handle.lib.cvae_big_steps_read(buf)
t = [v for v in buf if v]
d = [b - a for a, b in zip(t, t[1:])]
print(f"{len(t)} stamps; per step (stage = 5 steps; chunk = 25):")
for s0 in range(0, len(d), 5):
    print(f"  stage {s0 // 5:2d} (row {s0 // 5 % 5}): " + " ".join(f"{v:5d}" for v in d[s0:s0 + 5]) + f"   sum {sum(d[s0:s0 + 5])}")
