"""Stage timing of one bf16 conv instantiation (build: variant.sh ct conv_bf16.hip -DCONV_TIMING -DCONV_TIMING_KCH=32 -DCONV_TIMING_NCH=64
-DCONV_TIMING_H=32; run: CVAE_LIB=ab/ct.so python profiles/experiments/conv_timing.py).  Prints, per sampled workgroup and wave, the
cycles (s_memtime) spent waiting at the stage's first barrier, staging (LDS stores + issuing the next loads), waiting at the second
barrier and in the MFMA loop, summed over the stages, then the whole main loop and the epilogue."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd import synth  # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402
from critic_vae_amd.train import FusedTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
vae = VariationalAutoencoder(max_batch=B, seed=0, precision=PREC).to(dev)
tr = FusedTrainer(vae)
x = torch.rand(B, 3, 64, 64, device=dev)
pred = torch.rand(B, 1, device=dev)
eps = torch.randn(B, 32, device=dev)
for _ in range(5):
    tr.step(x, pred, eps)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 768)()
(vae.handle.lib.cvae_conv_dbg_read if PREC == "bf16" else vae.handle.lib.cvae_convf_dbg_read)(buf)
names = ["barrier1", "stage", "barrier2", "mfma"]
ST = 12 if PREC == "bf16" else 10          # the bf16 build also records s_memrealtime ticks (slot 10) for the in-kernel clock
for g in range(16):
    for w in range(4):
        t = list(buf[(g * 4 + w) * ST:(g * 4 + w) * ST + ST]) + [0, 0]
        if t[7] == 0:
            continue
        print(f"wg {64 * g:4d} wave {w}: " + "  ".join(f"{n} {t[i]}" for i, n in enumerate(names)) + f" (store_input {t[8]} store_w {t[9]})  mainloop {t[4]}  epilogue {t[5]}  stages {t[6]}  total {t[7]}  clock {t[7] / max(t[10], 1) * 0.1:.2f} GHz")
