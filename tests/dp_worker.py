"""2-rank gloo worker for tests/test_host.py::test_data_parallel_two_ranks_gloo (CPU only)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from critic_vae_amd import dp, layout as L, lib as cvlib, synth      # noqa: E402
from oracle import cvae_oracle as orc                                # noqa: E402

torch.set_num_threads(2)
world, rank, _ = dp.init("gloo")
assert world == 2
GLOBAL_B = 8
first, per = dp.shard_rows(GLOBAL_B, world, rank)
h = cvlib.Handle(64, per)


def shard_grad(r):
    f, n = dp.shard_rows(GLOBAL_B, world, r)
    x, pred, eps = (torch.from_numpy(a) for a in synth.make_batch(1234, 0, n, first_index=f))
    p = orc.to_torch(synth.make_params(0), requires_grad=True)
    orc.train_step(p, x, pred, eps)
    return L.ref_to_native(h.layout, h.param_total, {k: v.grad for k, v in p.items()})


# shards are disjoint slices of the global batch
xg, _, _ = synth.make_batch(1234, 0, GLOBAL_B)
xs, _, _ = synth.make_batch(1234, 0, per, first_index=first)
assert (xg[first:first + per] == xs).all()

mine = shard_grad(rank)
want = (shard_grad(0) + shard_grad(1)) / 2
got = dp.allreduce_mean_(mine.clone(), world)
err = (got - want).abs().max().item()
assert err <= 1e-7 * max(want.abs().max().item(), 1.0), err
assert dp.max_over_ranks(float(rank), torch.device("cpu")) == 1.0
print(f"DP_OK rank {rank} err {err:.2e}", flush=True)
dist.destroy_process_group()
