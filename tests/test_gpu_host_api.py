"""Host-side surface added around the C-ABI path: the workspace generation guard of the autograd route, the
stand-alone reparametrize / recon_samples of the reference API (vae_nets.py:21-29, 48-51), and the overlapped
uint8 frame feeder of the `-train` loop (vae.py:46-50, vae_utility.py:324-343)."""
import numpy as np
import pytest
import torch

from critic_vae_amd import synth
from critic_vae_amd.critic import Critic
from critic_vae_amd.feeder import FrameFeeder
from critic_vae_amd.nets import VariationalAutoencoder
from critic_vae_amd.train import FusedTrainer, train
from oracle import cvae_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vae(B, seed=0):
    vae = VariationalAutoencoder(max_batch=B, seed=seed).to(DEV)
    vae.load_reference_params(synth.make_params(seed))
    return vae


def _batch(B, step=0):
    return tuple(torch.from_numpy(a).to(DEV) for a in synth.make_batch(1234, step, B))


def test_backward_after_another_forward_raises():
    """The saved activations live in ONE workspace: a second forward (or an encoder / decoder call, or a fused step)
    between a forward and its backward must fail loudly, never return gradients of mixed batches."""
    B = 4
    vae = _vae(B)
    x, pred, eps = _batch(B)
    x2, pred2, eps2 = _batch(B, 1)
    for clobber in ("forward", "encoder", "decoder", "fused"):
        out = vae(x, pred, eps=eps)
        loss = vae.vae_loss(*out)["total_loss"]
        if clobber == "forward":
            vae(x2, pred2, eps=eps2)
        elif clobber == "encoder":
            vae.encoder(x2)
        elif clobber == "decoder":
            vae.decoder(out[1].detach(), pred2)
        else:
            FusedTrainer(vae, lr=0.0).step(x2, pred2, eps2)
        with pytest.raises(RuntimeError, match="overwritten"):
            loss.backward()
    # the plain sequence still works, and two losses of one forward may be summed
    vae.theta.grad = None
    out = vae(x, pred, eps=eps)
    (vae.vae_loss(*out)["total_loss"] + vae.vae_loss(*out)["total_loss"]).backward()
    g2 = vae.theta.grad.clone()
    vae.theta.grad = None
    out = vae(x, pred, eps=eps)
    vae.vae_loss(*out)["total_loss"].backward()
    torch.cuda.synchronize()
    assert torch.allclose(g2, 2 * vae.theta.grad, rtol=1e-6, atol=1e-9)


def test_reparametrize_and_recon_samples_follow_the_reference():
    """vae_nets.py:48-51 and :21-29: z = mu + randn_like(std) * exp(0.5 logvar) from torch's generator of the tensor's
    device; recon_samples = encoder once, six decoder passes of fresh samples.  Checked against the oracle's
    decoder on the very z the draw produced."""
    B = 3
    vae = _vae(B)
    vae.eval()
    x, pred, _ = _batch(B)
    mu, logvar = vae.encoder(x)
    torch.manual_seed(5)
    z = vae.reparametrize(mu, logvar)
    torch.manual_seed(5)
    std = torch.exp(0.5 * logvar)
    want = mu + torch.randn_like(std) * std
    assert torch.equal(z, want)
    torch.manual_seed(11)
    samples = vae.recon_samples(x, pred)
    assert len(samples) == 6 and all(s.shape == (B, 3, 64, 64) for s in samples)
    torch.manual_seed(11)
    p = orc.to_torch(synth.make_params(0))
    for s in samples:
        zi = mu + torch.randn_like(std) * std
        ref = orc.decoder(p, zi.cpu(), pred.cpu())
        assert (s.cpu() - ref).abs().max().item() < 1e-4
    assert not torch.equal(samples[0], samples[1])


def test_feeder_batches_equal_the_reference_preprocessing():
    """FrameFeeder (pinned double buffer, side-stream H2D, HIP uint8->CHW/255 + HIP critic) hands out exactly
    what vae.py:46-50 computes batch by batch — including a short tail batch and buffer re-use over many batches."""
    rng = np.random.default_rng(3)
    N, B = 70, 16
    frames = rng.integers(0, 256, size=(N, 64, 64, 3), dtype=np.uint8)
    vae = _vae(B)
    critic = Critic(handle=vae.handle).to(DEV)
    cp = synth.make_critic_params(0)
    critic.load_state_dict({k: torch.from_numpy(v) for k, v in cp.items()})
    order = [np.arange(N)[::-1][b:b + B].copy() for b in range(0, N, B)]
    feeder = FrameFeeder(frames, B, DEV, vae.handle, critic=critic)
    seen = 0
    for idx, (x, pred) in zip(order, feeder.batches(order)):
        want_x = orc.preprocess_frames(torch.from_numpy(frames[idx]))
        assert x.shape == (len(idx), 3, 64, 64) and torch.equal(x.cpu(), want_x)
        want_p = orc.critic_forward({k: torch.from_numpy(v) for k, v in cp.items()}, want_x)
        assert (pred.cpu() - want_p).abs().max().item() < 1e-5
        seen += len(idx)
    assert seen == N


def test_train_loop_on_uint8_frames_equals_the_float_route():
    """critic_vae_amd.train.train (vae.py:33-66) fed uint8 HWC frames through the feeder reaches the same losses and
    parameters as the reference-style route (host fp32 CHW dataset, synchronous copies) on the same shuffles."""
    rng = np.random.default_rng(4)
    N, B = 72, 16                                       # 4 full batches + a tail of 8
    frames = rng.integers(0, 256, size=(N, 64, 64, 3), dtype=np.uint8)
    as_float = [(frames[i].astype(np.float32) / 255.0).transpose(2, 0, 1)[None] for i in range(N)]
    res = []
    for dset in (frames, as_float):
        vae = _vae(B)
        critic = Critic(handle=vae.handle).to(DEV)
        critic.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_critic_params(0).items()})
        np.random.seed(9)
        torch.manual_seed(9)
        _, hist = train(vae, dset, critic if dset is frames else critic.evaluate, torch.device(DEV), epochs=2,
                        batch_size=B, log_n=B, log=lambda m: None)
        torch.cuda.synchronize()
        res.append((hist, vae.theta.detach().clone()))
    (h0, t0), (h1, t1) = res
    assert len(h0) == len(h1) == 10
    for (i0, r0), (i1, r1) in zip(h0, h1):
        assert i0 == i1 and all(abs(r0[k] - r1[k]) <= 1e-6 for k in r0)
    assert torch.equal(t0, t1)


def test_fused_trainer_fit_u8_runs_the_feeder():
    rng = np.random.default_rng(5)
    N, B = 40, 16
    frames = rng.integers(0, 256, size=(N, 64, 64, 3), dtype=np.uint8)
    vae = _vae(B)
    critic = Critic(handle=vae.handle).to(DEV)
    critic.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_critic_params(0).items()})
    tr = FusedTrainer(vae)
    before = vae.theta.detach().clone()
    scal = tr.fit_u8(frames, critic, B, epochs=1, generator=torch.Generator(device=DEV).manual_seed(1))
    torch.cuda.synchronize()
    assert tr.step_count == 3 and torch.isfinite(scal[:3]).all() and not torch.equal(before, vae.theta.detach())


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_side_stream_weight_gradients_are_bit_identical(precision):
    """cvae_config.overlap_wgrad (opt-in; bench.py times it as config2.side_stream_wgrad) moves the weight-gradient kernels and their slab reductions to the
    library's side stream: same kernels, same arithmetic, so two steps give bit-identical gradients, parameters and scalars."""
    import torch
    from critic_vae_amd import synth
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer
    dev = torch.device("cuda:0")
    B = 24
    outs = []
    for overlap in (False, True):
        vae = VariationalAutoencoder(max_batch=B, seed=0, precision=precision, overlap_wgrad=overlap).to(dev)
        tr = FusedTrainer(vae)
        sc = []
        for s in range(3):
            x, pred, eps = (torch.from_numpy(a).to(dev) for a in synth.make_batch(77, s, B))
            sc.append(tr.step(x, pred, eps).clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(sc), tr.grads.clone(), vae.theta.data.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
