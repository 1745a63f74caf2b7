"""The N>1 path of FusedTrainer on a real GPU: two ranks on the one device of the test box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_bucketed_overlap_and_single_allreduce():
    env = dict(os.environ, CVAE_DIST_BACKEND="gloo", CVAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                        "--nnodes=1", "--nproc-per-node=2",            # --standalone: torchrun picks a free rendezvous port
                        os.path.join(ROOT, "tests", "dp_gpu_worker.py")], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DP_GPU_OK rank 0" in r.stdout and "DP_GPU_OK rank 1" in r.stdout


def test_bf16_gradient_pack_is_rne_and_exact_back():
    """cvae_grads_to_bf16 == torch's fp32 -> bfloat16 cast (round to nearest even) bit for bit — NaN, infinities, zeros,
    subnormals and exact midpoints included — also on a bucket slice; cvae_grads_from_bf16 is exact."""
    import torch
    from critic_vae_amd.lib import Handle
    dev = torch.device("cuda:0")
    h = Handle(64, 4)
    g = torch.Generator(device=dev).manual_seed(3)
    n = 1 << 16
    v = torch.randn(n, device=dev, generator=g) * torch.exp(torch.randn(n, device=dev, generator=g) * 8)
    special = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e-40, -1e-40, 3.0e38, -3.0e38,
                            1.00390625, 1.01171875, -1.00390625, 65280.0 * 2.0 ** 112, 1.0 + 2.0 ** -8 + 2.0 ** -20], device=dev)
    v[:special.numel()] = special                      # 1.00390625 = 1 + 2^-8: an exact midpoint (ties to even -> 1.0)
    out = torch.empty(n, dtype=torch.bfloat16, device=dev)
    h.grads_to_bf16(v, out)
    want = v.to(torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16)[~torch.isnan(v)], want.view(torch.int16)[~torch.isnan(v)])
    assert torch.isnan(out.float()[torch.isnan(v)]).all()
    back = torch.empty(n, device=dev)
    h.grads_from_bf16(out, back)
    torch.cuda.synchronize()
    ok = ~torch.isnan(v)
    assert torch.equal(back[ok], want.float()[ok])
    # a bucket slice (offset and length multiples of 64, as cvae_grad_bucket returns them)
    out2 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    h.grads_to_bf16(v[4096:12288], out2[4096:12288])
    torch.cuda.synchronize()
    assert torch.equal(out2[4096:12288].view(torch.int16)[ok[4096:12288]], want[4096:12288].view(torch.int16)[ok[4096:12288]])
    assert not out2[:4096].any() and not out2[12288:].any()
    h.grads_to_bf16(v[:0], out2[:0])                   # empty range: a no-op, not an invalid launch


@pytest.mark.parametrize("precision,side", [("f32", False), ("bf16", False), ("bf16", True)])      # side: weight gradients on the side stream (joined per phase)
def test_phased_backward_equals_single_call(precision, side):
    import torch
    from critic_vae_amd import synth
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer
    dev = torch.device("cuda:0")
    B = 5
    vae = VariationalAutoencoder(max_batch=B, seed=0, precision=precision, overlap_wgrad=side).to(dev)
    tr = FusedTrainer(vae)
    x, pred, eps = (torch.from_numpy(a).to(dev) for a in synth.make_batch(1234, 0, B))
    h, theta = vae.handle, vae.theta.data
    h.forward(B, x, pred, eps, theta, vae.bn_state, tr.mu, tr.logvar, tr.recon, tr.ws, train=True)
    h.loss(B, x, tr.mu, tr.logvar, tr.recon, tr.ws, tr.scalars, tr.d_recon, tr.d_mu, tr.d_logvar)
    h.backward(B, x, pred, eps, theta, tr.logvar, tr.recon, tr.d_recon, tr.d_mu, tr.d_logvar, tr.ws, tr.grads)
    whole = tr.grads.clone()
    used = torch.zeros_like(whole, dtype=torch.bool)            # everything except the alignment padding
    for off, n in h.layout.values():
        used[off:off + n] = True
    tr.grads.fill_(float("nan"))
    covered = torch.zeros_like(used)
    for ph in range(3):
        h.backward_phase(ph, B, x, pred, eps, theta, tr.logvar, tr.recon, tr.d_recon, tr.d_mu, tr.d_logvar, tr.ws, tr.grads)
        off, n = h.grad_bucket(ph)
        torch.cuda.synchronize()
        assert torch.isfinite(tr.grads[off:off + n][used[off:off + n]]).all()       # the bucket is complete after its phase
        covered[off:off + n] = True
    assert covered.all()
    assert torch.equal(tr.grads[used], whole[used])


def test_bucketed_allreduce_path_on_rccl_single_rank():
    """The exact N>1 code path (phased backward, async all_reduce per bucket, wait, Adam with 1/N) on
    the real nccl (= RCCL) backend with one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_nccl_worker.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DP_NCCL_OK" in r.stdout


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` from plain python (no torchrun environment) starts the ranks itself; on the
    one-GPU test box both ranks share device 0 and gloo carries the collectives."""
    import json
    env = dict(os.environ, CVAE_DIST_BACKEND="gloo", CVAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "16"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    d = res["config"]["distributed"]
    assert res["n_gpus"] == 2 and d["ranks_counted_by_allreduce"] == 2 and d["world_size"] == 2
    assert d["backend"] == "gloo" and d["rank_devices"] == [0, 0] and d["allreduce_exposed_us"] is not None
    assert res["config"]["global_batch"] == 32 and res["value"] > 0 and res["config"]["loss_finite"]


def test_bench_multi_gpu_line_carries_config4_and_config5():
    """The default N > 1 bench run (what the driver starts on the 8-GPU node) also times BASELINE configs[3] and configs[4]
    per GPU, each with the three overlapped gradient buckets and with the single all-reduce after backward.  Rehearsed here
    with two ranks on the one GPU of the test box (gloo carries the collectives)."""
    import json
    env = dict(os.environ, CVAE_DIST_BACKEND="gloo", CVAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 512 and res["config"]["preset"] == "config1"
    for key, gb in (("config4", 4096), ("config5", 2048)):
        c = res["config"][key]
        assert c["global_batch"] == gb and c["value"] > 0 and c["loss_finite"]
        assert c["grad_allreduce"].startswith("3 buckets") and c["allreduce_exposed_us"] is not None
        single = c["allreduce_modes"]["single"]
        assert single["grad_allreduce"].startswith("single") and single["allreduce_exposed_us"] is not None and single["value"] > 0
        assert c["distributed"]["ranks_counted_by_allreduce"] == 2


def test_two_handles_are_independent():
    """One handle per configuration, no shared state: two handles of different frame size / precision used
    alternately give bit-identical results to each one used alone."""
    import torch
    from critic_vae_amd import synth
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer
    dev = torch.device("cuda:0")

    def run(specs, interleave):
        trs = [FusedTrainer(VariationalAutoencoder(width=w, max_batch=4, seed=0, precision=pr).to(dev)) for w, pr in specs]
        outs = [[] for _ in specs]
        order = [(i, s) for s in range(2) for i in range(len(specs))] if interleave else \
                [(i, s) for i in range(len(specs)) for s in range(2)]
        for i, s in order:
            x, pred, eps = (torch.from_numpy(a).to(dev) for a in synth.make_batch(1234, s, 4, specs[i][0]))
            outs[i].append(trs[i].step(x, pred, eps).clone())
        torch.cuda.synchronize()
        return [torch.stack(o) for o in outs], [t.vae.theta.data.clone() for t in trs]

    specs = [(64, "f32"), (128, "f32"), (64, "bf16")]
    a_s, a_t = run(specs, interleave=True)
    b_s, b_t = run(specs, interleave=False)
    for i in range(len(specs)):
        assert torch.isfinite(a_s[i][:, :3]).all()
        assert torch.equal(a_s[i], b_s[i]) and torch.equal(a_t[i], b_t[i]), specs[i]
