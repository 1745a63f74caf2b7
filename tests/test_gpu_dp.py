"""The N>1 path of FusedTrainer on a real GPU: two ranks on the one device of the test box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_bucketed_overlap_and_single_allreduce():
    env = dict(os.environ, CVAE_DIST_BACKEND="gloo", CVAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(ROOT, "tests", "dp_gpu_worker.py")], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DP_GPU_OK rank 0" in r.stdout and "DP_GPU_OK rank 1" in r.stdout


def test_phased_backward_equals_single_call():
    import torch
    from critic_vae_amd import synth
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer
    dev = torch.device("cuda:0")
    B = 5
    vae = VariationalAutoencoder(max_batch=B, seed=0).to(dev)
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
