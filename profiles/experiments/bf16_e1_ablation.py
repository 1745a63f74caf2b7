"""Where does the bf16-mode error of the FIRST conv's weight gradient (rel L2 0.10 at 64x64 / 0.16 at 128x128, the worst
tensor of tests/test_gpu_bf16.py::test_bf16_full_size_configs) come from?   python profiles/experiments/bf16_e1_ablation.py [B] [W]

dW1 = wgrad(x, dy0), dy0 = BatchNorm/pool/ReLU backward of d_a0, d_a0 = input gradient of E2.  One bf16 HIP step gives
the d_a0 the bf16 chain really produced (read from the workspace); the fp32 oracle gives the exact one.  Block 0's
backward is then re-run in fp32 (torch autograd on the CPU) from either d_a0, with the kernel's own roundings (x -> bf16,
dy0 -> bf16) switched on one at a time."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd import layout as L, synth                     # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder            # noqa: E402
from critic_vae_amd.train import FusedTrainer                     # noqa: E402
from oracle import cvae_oracle as orc                             # noqa: E402  (checker only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
bf = lambda t: t.to(torch.bfloat16).float()                       # noqa: E731


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


params = synth.make_params(0, W)
x, pred, eps = (torch.from_numpy(v) for v in synth.make_batch(1234, 0, B, W))
p = orc.to_torch(params, requires_grad=True)
taps = {}
orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p), taps=taps)
dW_ref = p["encoder.model.0.weight"].grad.clone()
d_a0_ref = taps["enc_a0"].grad.clone()

vae = VariationalAutoencoder(width=W, max_batch=B, seed=0, precision="bf16").to(dev)
tr = FusedTrainer(vae, lr=0.0)
tr.step(x.to(dev), pred.to(dev), eps.to(dev))
torch.cuda.synchronize()
h = vae.handle
dW_hip = L.native_to_ref(h.layout, tr.grads.cpu())["encoder.model.0.weight"]
off = h.lib.cvae_ws_offset(h.h, B, b"d_a0")
n = B * (W // 2) * (W // 2) * 32
d_a0_hip = tr.ws.view(torch.bfloat16)[2 * off:2 * off + n].float().view(B, W // 2, W // 2, 32).permute(0, 3, 1, 2).contiguous().cpu()


def block0_dW(d_a0, round_x=False, round_dy=False, round_y_for_pool=False):
    w1 = p["encoder.model.0.weight"].detach().clone().requires_grad_(True)
    b1 = p["encoder.model.0.bias"].detach()
    y = F.conv2d(bf(x) if round_x else x, bf(w1) if round_x else w1, b1, padding=2)
    if round_y_for_pool:
        y = y + (bf(y) - y).detach()          # the forward pools bf16-rounded values; the gradient passes straight through
    y.retain_grad()
    nrm = F.batch_norm(y, None, None, p["encoder.model.1.weight"].detach(), p["encoder.model.1.bias"].detach(), training=True, eps=1e-5)
    a = torch.relu(F.max_pool2d(nrm, 2))
    a.backward(d_a0)
    dy = bf(y.grad) if round_dy else y.grad
    return torch.nn.grad.conv2d_weight(bf(x) if round_x else x, w1.shape, dy, padding=2)


rows = [
    ("HIP bf16 step (what the test measures)", rel(dW_hip, dW_ref)),
    ("d_a0 of the bf16 chain vs fp32 d_a0 (input of block 0's backward)", rel(d_a0_hip, d_a0_ref)),
    ("fp32 block-0 backward from the EXACT d_a0 (sanity: 0)", rel(block0_dW(d_a0_ref), dW_ref)),
    ("  + d_a0 stored as bf16 only", rel(block0_dW(bf(d_a0_ref)), dW_ref)),
    ("  + x and W1 rounded to bf16 in conv / wgrad only", rel(block0_dW(d_a0_ref, round_x=True), dW_ref)),
    ("  + dy0 rounded to bf16 (the wgrad MFMA operand) only", rel(block0_dW(d_a0_ref, round_dy=True), dW_ref)),
    ("  + forward pools bf16-rounded y only (argmax flips)", rel(block0_dW(d_a0_ref, round_y_for_pool=True), dW_ref)),
    ("  all three local roundings, exact d_a0", rel(block0_dW(d_a0_ref, True, True, True), dW_ref)),
    ("fp32 block-0 backward from the bf16 chain's d_a0", rel(block0_dW(d_a0_hip), dW_ref)),
    ("  + all three local roundings (= an emulation of the HIP kernel)", rel(block0_dW(d_a0_hip, True, True, True), dW_ref)),
]
print(f"bf16 E1 weight-gradient ablation, B={B}, {W}x{W}: relative L2 error of dW1 against the fp32 oracle")
for name, v in rows:
    print(f"  {v:9.4f}  {name}")
g = dW_ref.double()
terms = torch.nn.grad.conv2d_weight(x.abs(), dW_ref.shape, taps["enc_y0"].grad.abs(), padding=2).double()
print(f"  cancellation in the fp32 wgrad sum: |sum x*dy| / sum |x*dy| = {float(g.norm() / terms.norm()):.2e} "
      f"(BatchNorm's backward removes the mean and the xhat component of dy0: the sum is a small remainder of large terms)")
