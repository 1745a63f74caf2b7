"""Phase timing of one instantiation of the persistent bf16 conv kernel (build: variant.sh pt conv_bf16_ps.hip -DPS_TIMING -DPS_T_KCH=32
-DPS_T_NCH=64 -DPS_T_H=32; run: CVAE_LIB=ab/pt.so python profiles/experiments/ps_timing.py [B]): cycles per ITEM (tile pair x
channel block) of sampled workgroups, wave 0: barrier 1 | staging (LDS stores + next loads issued) | drain | barrier 2 | MFMA | epilogue."""
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
buf = (ctypes.c_longlong * 192)()
vae.handle.lib.cvae_ps_dbg_read(buf)
names = ["barrier1", "staging", "drain", "barrier2", "mfma", "epilogue"]
starts = [buf[g * 12 + 9] for g in range(16) if buf[g * 12]]
entry = [buf[g * 12 + 10] for g in range(16) if buf[g * 12]]
exits = [buf[g * 12 + 11] for g in range(16) if buf[g * 12]]
print(f"kernel entry -> loop start {[(a - b) / 100.0 for a, b in zip(starts, entry)][:6]} us; first entry -> last exit {(max(exits) - min(entry)) / 100.0:.1f} us; loop end -> exit {[(e - s - buf[g * 12 + 8]) / 100.0 for g, (e, s) in enumerate(zip(exits, starts))][:6]} us")
print("loop start of the sampled workgroups relative to the earliest (us):", [round((v - min(starts)) / 100.0, 1) for v in starts])
for g in range(16):
    t = list(buf[g * 12:g * 12 + 12])
    if t[0] == 0:
        continue
    n = t[0]
    print(f"wg {32 * g:4d}: {n} items; per item: " + "  ".join(f"{nm} {t[i + 1] // n}" for i, nm in enumerate(names)) +
          f"  sum {sum(t[1:7]) // n}  whole loop {t[7] // n}  in-kernel clock {t[7] / max(t[8], 1) * 0.1:.2f} GHz  ({t[8] / 100.0:.1f} us)")
