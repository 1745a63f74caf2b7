import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from critic_vae_amd import synth
from critic_vae_amd.nets import VariationalAutoencoder
from critic_vae_amd.train import FusedTrainer
B = 32
def model():
    v = VariationalAutoencoder(max_batch=B, seed=0).cuda(); v.load_reference_params(synth.make_params(0)); return v
a, b = model(), model()
x, pred, eps = (torch.from_numpy(t).cuda() for t in synth.make_batch(1234, 0, B))
opt = torch.optim.Adam(a.parameters(), lr=5e-5)
tr = FusedTrainer(b)
lay = a.handle.layout
def where(i):
    for k,(off,n) in lay.items():
        if off <= i < off+n: return k, i-off
    return 'pad', i
for step in range(3):
    opt.zero_grad()
    out = a(x, pred, eps=eps); l = a.vae_loss(*out); l['total_loss'].backward()
    ga = a.theta.grad.clone()
    opt.step()
    tr.step(x, pred, eps)
    torch.cuda.synchronize()
    dg = (ga - tr.grads).abs(); dp = (a.theta - b.theta).abs()
    ig, ip = int(dg.argmax()), int(dp.argmax())
    print(step, 'grad diff', dg.max().item(), where(ig), 'param diff', dp.max().item(), where(ip),
          'g there', ga[ip].item(), tr.grads[ip].item(), 'm', opt.state[a.theta]['exp_avg'][ip].item(), tr.m[ip].item(),
          'v', opt.state[a.theta]['exp_avg_sq'][ip].item(), tr.v[ip].item())
