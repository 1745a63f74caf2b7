"""Distribution of max-pool / ReLU decision flips against the oracle and of the value deviation behind them, per precision mode
(VERDICT round 4, weak 1a): python profiles/experiments/tie_gap_study.py > profiles/r05_tie_gap_study.txt  (GPU box; ~2 min of CPU oracle).
For every (mode, frame size, data seed): flips as [(decision, count, largest distance to a tie)], E = max |n_hip - n_orc| per layer, and the
largest flip gap in units of 2 E (<= 1 by construction of a round-off flip, tests/decisions.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from critic_vae_amd import synth  # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402
from oracle import cvae_oracle as orc  # noqa: E402
from decisions import hip_decisions, oracle_decisions, flips, value_deviation  # noqa: E402

dev = torch.device("cuda:0")
worst = {}
for W, B in ((64, 32), (128, 3)):
    for dseed in (1234, 7, 11, 21, 33):
        x, pred, eps = (torch.from_numpy(v) for v in synth.make_batch(dseed, 0, B, W))
        p = orc.to_torch(synth.make_params(0, W), requires_grad=True)
        taps = {}
        o = orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p), taps=taps)
        if not torch.isfinite(o["total_loss"]):
            print(f"W={W} B={B} seed={dseed}: oracle loss not finite, skipped")
            continue
        d_orc = oracle_decisions(taps)
        for mode in ("f32", "bf16x9", "bf16x6"):
            vae = VariationalAutoencoder(width=W, max_batch=B, seed=0, precision=mode).to(dev)
            vae.load_reference_params(synth.make_params(0, W))
            out = vae(x.to(dev), pred.to(dev), eps=eps.to(dev))
            vae.vae_loss(*out)["total_loss"].backward()
            torch.cuda.synchronize()
            fl = flips(hip_decisions(vae, B), d_orc, taps)
            devn = value_deviation(vae, B, taps)
            gmax = max([g for _, _, g in fl], default=0.0)
            ratio = max([g / (2 * devn[("enc" if "enc" in k or k.startswith("pool") else "dec") + k[-1]] + 1e-30) for k, _, g in fl], default=0.0)
            worst[mode] = max(worst.get(mode, 0.0), gmax)
            e_enc = " ".join("%.1e" % devn["enc%d" % l] for l in range(4))
            e_dec = " ".join("%.1e" % devn["dec%d" % i] for i in range(4))
            print(f"W={W} B={B} seed={dseed} {mode:7s}: {sum(f[1] for f in fl):3d} flips, largest gap {gmax:.3e} ({ratio:.2f} of 2E)  E enc [{e_enc}] dec [{e_dec}]  {fl}")
print("largest flipped-unit gap per mode:", {k: f"{v:.3e}" for k, v in worst.items()})
