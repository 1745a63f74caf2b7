"""Experiment: does replaying the whole step as one HIP graph beat stream launches?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from critic_vae_amd.nets import VariationalAutoencoder
from critic_vae_amd.train import FusedTrainer

B = 256
dev = torch.device("cuda:0")
vae = VariationalAutoencoder(max_batch=B, seed=0).to(dev)
tr = FusedTrainer(vae)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(B, 3, 64, 64, device=dev, generator=g); p = torch.rand(B, 1, device=dev, generator=g)
e = torch.randn(B, 32, device=dev, generator=g)
for _ in range(5):
    tr.step(x, p, e)
torch.cuda.synchronize()
def timeit(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("stream ms/step", timeit(lambda: tr.step(x, p, e)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.step(x, p, e)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        tr.step(x, p, e)
    print("graph ms/step", timeit(gr.replay), flush=True)
    print("stream again ", timeit(lambda: tr.step(x, p, e)), flush=True)
except Exception as ex:
    print("capture failed:", repr(ex)[:500])
