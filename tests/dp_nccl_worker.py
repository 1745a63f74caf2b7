"""Single-rank RCCL worker for tests/test_gpu_dp.py: the bucketed, overlapped all-reduce path of
FusedTrainer (phased backward + async all_reduce on slices of the flat gradient + wait) on the real
`nccl` backend.  One rank only — RCCL refuses two ranks on one device — so the sums are identities,
but every call the 8-GPU run makes is made."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from critic_vae_amd import synth                          # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder    # noqa: E402
from critic_vae_amd.train import FusedTrainer             # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
import tempfile                                           # noqa: E402
_store = os.path.join(tempfile.mkdtemp(prefix="cvae_dp_"), "store")      # file rendezvous: no TCP port to collide on
dist.init_process_group(backend="nccl", init_method="file://" + _store, rank=0, world_size=1, device_id=dev)
B = 8
x, pred, eps = (torch.from_numpy(a).to(dev) for a in synth.make_batch(1234, 0, B))

ref = FusedTrainer(VariationalAutoencoder(max_batch=B, seed=0).to(dev))
ref.step(x, pred, eps)

vae = VariationalAutoencoder(max_batch=B, seed=0).to(dev)
tr = FusedTrainer(vae)
tr.world_size, tr.overlap = 2, True            # take the N>1 branch: three async bucket all-reduces over RCCL
for _ in range(3):
    scal = tr.step(x, pred, eps)
torch.cuda.synchronize()
assert torch.isfinite(scal[:3]).all()
# first step's gradient path is identical to the single-GPU one (sum over one rank); Adam saw grad/2
tr2 = FusedTrainer(VariationalAutoencoder(max_batch=B, seed=0).to(dev))
tr2.world_size, tr2.overlap = 2, True
tr2.step(x, pred, eps)
torch.cuda.synchronize()
assert torch.equal(tr2.grads, ref.grads), "bucketed RCCL path changed the gradient"
# the optional bf16 wire format over RCCL: one rank, so the reduced gradient is exactly the bf16 rounding of the fp32 one
tr3 = FusedTrainer(VariationalAutoencoder(max_batch=B, seed=0).to(dev), reduce_dtype="bf16", sync=False)
tr3.world_size, tr3.overlap = 2, True
tr3.grads16 = torch.empty(tr3.grads.numel(), dtype=torch.bfloat16, device=dev)
tr3.step(x, pred, eps)
torch.cuda.synchronize()
assert torch.equal(tr3.grads, ref.grads.to(torch.bfloat16).float()), "bf16 bucket all-reduce over RCCL is not RNE(g)"
t = torch.arange(1000, device=dev, dtype=torch.float32)
w = dist.all_reduce(t[100:600], async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(t, torch.arange(1000, device=dev, dtype=torch.float32))
print("DP_NCCL_OK", flush=True)
dist.destroy_process_group()
