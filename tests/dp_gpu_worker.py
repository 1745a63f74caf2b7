"""2-rank worker for tests/test_gpu_dp.py: the data-parallel FusedTrainer step on the GPU (both ranks
on device 0, gloo carrying the collectives — RCCL refuses two ranks on one device), with the bucketed
all-reduce overlapped with backward and with the single all-reduce, against the contract of SURVEY.md
§8e: reduced gradient == mean of the per-shard single-rank gradients, then one identical Adam step."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from critic_vae_amd import dp, synth                      # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder    # noqa: E402
from critic_vae_amd.train import FusedTrainer             # noqa: E402

world, rank, local = dp.init()
assert world == 2
dev = torch.device("cuda", dp.device_index(local))
torch.cuda.set_device(dev)
GLOBAL_B = 16
first, per = dp.shard_rows(GLOBAL_B, world, rank)


def batch(r):
    f, n = dp.shard_rows(GLOBAL_B, world, r)
    return tuple(torch.from_numpy(a).to(dev) for a in synth.make_batch(1234, 0, n, first_index=f))


def single_rank_grad(r):
    vae = VariationalAutoencoder(max_batch=per, seed=0).to(dev)
    tr = FusedTrainer(vae)                                   # world 1: grads of this shard only
    tr.step(*batch(r))
    return tr.grads.clone()


# replicas built from DIFFERENT seeds (seed=None draws from each process's own RNG in real use) must be
# identical once a data-parallel trainer exists: FusedTrainer broadcasts rank 0's parameters, BatchNorm
# statistics and Adam state (torch DDP does the same)
vae_d = VariationalAutoencoder(max_batch=per, seed=100 + rank).to(dev)
before = vae_d.theta.data.clone()
tr_d = FusedTrainer(vae_d, world_size=world)
r0 = vae_d.theta.data.clone()
dist.broadcast(r0, src=0)
assert torch.equal(r0, vae_d.theta.data), "replicas differ after FusedTrainer construction"
assert rank == 0 or not torch.equal(before, vae_d.theta.data)
tr_d.step(*batch(rank))
r0 = vae_d.theta.data.clone()
dist.broadcast(r0, src=0)
assert torch.equal(r0, vae_d.theta.data), "replicas diverged after one step"
del vae_d, tr_d

g_shard = [single_rank_grad(0), single_rank_grad(1)]
want_grad = g_shard[0] + g_shard[1]                           # FusedTrainer keeps the SUM; 1/N is folded into Adam
results = {}
for overlap in (True, False):
    vae = VariationalAutoencoder(max_batch=per, seed=0).to(dev)
    tr = FusedTrainer(vae, world_size=world, overlap=overlap)
    assert tr.overlap == overlap
    tr.step(*batch(rank))
    torch.cuda.synchronize()
    err = (tr.grads - want_grad).abs().max().item()
    assert err <= 2e-6 * max(want_grad.abs().max().item(), 1.0), (overlap, err)
    results[overlap] = vae.theta.data.clone()
    # every rank holds the same parameters after the step
    other = results[overlap].clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, results[overlap]), "ranks diverged"
assert torch.equal(results[True], results[False]), "bucketed and single all-reduce differ"
# optional bf16 wire format (SURVEY 8e "bf16 optional"): each rank sends p_r = RNE_bf16(g_r) (cvae_grads_to_bf16 is torch's
# own fp32 -> bfloat16 cast bit for bit: tests/test_gpu_dp.py::test_bf16_gradient_pack_is_rne_and_exact_back) and the
# backend returns p_0 + p_1 in bf16.  The reference value is therefore computed HERE from the two packed shards: the
# exact sum of two bf16 numbers, and the result must be that sum rounded FAITHFULLY to bf16 (|error| < 1 ulp =
# 2^-7 of the result's binade) whatever rounding the backend's bf16 addition uses — round-to-nearest (<= 1/2 ulp) and
# truncation both satisfy it; which one this backend does is printed.  Every rank ends with the same parameters, and
# the bucketed / single variants agree to the bit (same per-element operations).
packed = [g.to(torch.bfloat16) for g in g_shard]
exact = packed[0].double() + packed[1].double()
rne = (packed[0].float() + packed[1].float()).to(torch.bfloat16).float()      # fp32 sum of two bf16 values, then one RNE
res16 = {}
for overlap in (True, False):
    vae = VariationalAutoencoder(max_batch=per, seed=0).to(dev)
    tr16 = FusedTrainer(vae, world_size=world, overlap=overlap, reduce_dtype="bf16")
    tr16.step(*batch(rank))
    torch.cuda.synchronize()
    got = tr16.grads
    assert torch.equal(got, got.to(torch.bfloat16).float()), "the reduced gradient must be bf16-representable"
    d = (got.double() - exact).abs()
    ulp = 2.0 ** (torch.floor(torch.log2(torch.maximum(got.double().abs(), exact.abs()).clamp_min(1e-300))) - 7)
    assert (d < ulp + 1e-300).all(), (overlap, (d / ulp).max().item())
    if rank == 0 and overlap:
        print(f"bf16 wire: backend {dist.get_backend()} sum == round-to-nearest of the exact sum on "
              f"{(got == rne).float().mean().item() * 100:.3f} % of the elements; max error {(d / ulp).max().item():.3f} ulp", flush=True)
    res16[overlap] = vae.theta.data.clone()
    other = res16[overlap].clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, res16[overlap]), "ranks diverged (bf16 reduce)"
assert torch.equal(res16[True], res16[False])
# buckets tile the flat buffer exactly
b = sorted(tr.buckets)
assert b[0][0] == 0 and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(2)) and b[2][0] + b[2][1] == tr.grads.numel()
print(f"DP_GPU_OK rank {rank} err {err:.2e}", flush=True)
dist.destroy_process_group()
