"""The `-train` path of the reference (vae.py:33-66, 154-163) on synthetic data.

Two drivers over the same C-ABI kernels:

  * train(autoencoder, dset, critic_fn)  — the reference loop, line for line (torch.optim.Adam over
    autoencoder.parameters(), np.random.shuffle per epoch, short tail batch kept, log every log_n).
  * FusedTrainer                          — the same step without autograd/optimizer objects: direct
    cvae_forward -> cvae_loss -> cvae_backward -> (RCCL all-reduce of the flat gradient) ->
    cvae_adam_step, all asynchronous on one stream.  This is what bench.py times.

    python -m critic_vae_amd.train -train --synthetic 1024 --batch 32 --epochs 1
"""
import argparse
import os
import time

import numpy as np
import torch

from . import params as P
from . import synth
from .nets import VariationalAutoencoder


def _reference_batches(dset, epoch_indices, starts, batch_size, device, critic_fn):
    """vae.py:46-50 as the reference does it: gather, fp32 copy to the device, critic — all synchronous."""
    for b in starts:
        images = torch.from_numpy(dset[epoch_indices[b:b + batch_size]]).to(device=device, dtype=torch.float32)
        yield images, critic_fn(images)


def train(autoencoder, dset, critic_fn, device, epochs=P.epochs, batch_size=P.batch_size, lr=P.lr,
          log_n=None, log=print):
    """vae.py:33-66.  `dset`: list/array of (1,3,w,w) or (3,w,w) float frames in [0,1] — the reference's own
    host-side format: every batch is converted and copied synchronously, as vae.py:46-48 does — OR one uint8
    array (N,w,w,3), the frames as the environment delivers them: batches then come through FrameFeeder
    (pinned double buffer, H2D on a side stream, uint8 -> fp32 CHW/255 and the critic on the GPU, batch i+1
    in flight under step i).  `critic_fn(images) -> (B,1)` stands in for critic.evaluate (vae.py:50); with
    uint8 input it may also be a critic_vae_amd.critic.Critic."""
    u8 = isinstance(dset, np.ndarray) and dset.dtype == np.uint8
    if not u8:
        dset = np.stack(dset).squeeze()
    opt = torch.optim.Adam(autoencoder.parameters(), lr=lr)
    num_samples = dset.shape[0]
    log_n = log_n if log_n is not None else batch_size * 30
    history = []
    feeder = None
    if u8:
        from .feeder import FrameFeeder
        feeder = FrameFeeder(dset, batch_size, device, autoencoder.handle, critic=critic_fn)
    for ep in range(epochs):
        epoch_indices = np.arange(num_samples)
        np.random.shuffle(epoch_indices)
        starts = range(0, num_samples, batch_size)                              # tail batch is kept (vae.py:44-46)
        if feeder is not None:
            stream = feeder.batches([epoch_indices[b:b + batch_size] for b in starts])
        else:
            stream = _reference_batches(dset, epoch_indices, starts, batch_size, device, critic_fn)
        for batch_i, (images, preds) in zip(starts, stream):
            opt.zero_grad()
            out = autoencoder(images, preds)
            losses = autoencoder.vae_loss(out[0], out[1], out[2], out[3])
            losses["total_loss"].backward()
            opt.step()
            if batch_i % log_n == 0:
                rec = {k: float(v.item()) for k, v in losses.items()}
                history.append((num_samples * ep + batch_i + 1, rec))
                log(f"    ep:{ep}, imgs:{num_samples * ep + (batch_i + 1)} {rec}")
    return autoencoder, history


class FusedTrainer:
    """One training step = forward + loss + backward + all-reduce + Adam on flat buffers."""

    def __init__(self, vae, lr=P.lr, betas=P.adam_betas, eps=P.adam_eps, process_group=None, world_size=1,
                 overlap=None, reduce_dtype=None, sync=True):
        """Construction with world_size > 1 is a COLLECTIVE (sync_replicas: five broadcasts from rank 0) unless
        sync=False.
        overlap: all-reduce the gradient in three buckets while backward still runs (default for
        world_size > 1; CVAE_DP_OVERLAP=0 or overlap=False = one all-reduce after backward).
        reduce_dtype: "f32" (default; the contract of SURVEY 8e: reduced gradient == mean of the shard gradients
        within 1e-4) or "bf16" (CVAE_DP_REDUCE=bf16): the wire format of the all-reduce is bf16 — half the bytes,
        the summed gradient carries a relative rounding of 2^-9 per rank, optimizer state stays fp32."""
        self.vae = vae
        self.h = vae.handle
        if overlap is None:
            overlap = os.environ.get("CVAE_DP_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and world_size > 1
        if reduce_dtype is None:
            reduce_dtype = os.environ.get("CVAE_DP_REDUCE", "f32")
        if reduce_dtype not in ("f32", "bf16"):
            raise ValueError(f"reduce_dtype {reduce_dtype!r}: 'f32' or 'bf16'")
        self.reduce_dtype = reduce_dtype
        self.buckets = [self.h.grad_bucket(ph) for ph in range(3)]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.world_size, self.pg = world_size, process_group
        dev = vae.theta.device
        n = vae.theta.numel()
        self.grads = torch.zeros(n, device=dev)
        self.grads16 = torch.empty(n, dtype=torch.bfloat16, device=dev) if (reduce_dtype == "bf16" and world_size > 1) else None
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.step_count = 0
        B = vae.max_batch
        self.mu = torch.empty(B, P.latent_dim, device=dev)
        self.logvar = torch.empty_like(self.mu)
        self.recon = torch.empty(B, P.ch, vae.width, vae.width, device=dev)
        self.d_recon = torch.empty_like(self.recon)
        self.d_mu = torch.empty_like(self.mu)
        self.d_logvar = torch.empty_like(self.mu)
        self.scalars = torch.empty(16, device=dev)
        self.ws = vae._workspace(B)
        self.exposed_ms = []              # measure_exposed: device time the compute stream spent waiting for the all-reduce
        self.measure_exposed = False
        self._ev = None
        if world_size > 1 and sync:
            if not torch.distributed.is_initialized():
                raise RuntimeError("FusedTrainer(world_size > 1) broadcasts rank 0's replica at construction (a COLLECTIVE: "
                                   "every rank must construct its trainer, in the same order): call "
                                   "torch.distributed.init_process_group / critic_vae_amd.dp.init() first, or pass "
                                   "sync=False and call sync_replicas() yourself")
            self.sync_replicas()

    def sync_replicas(self, src=0):
        """Every rank starts from rank `src`'s replica: parameters, BatchNorm running statistics and the Adam
        state are broadcast (torch DDP does the same at construction) — ranks built with different seeds,
        or one rank restored from a checkpoint, would otherwise train different models on the averaged
        gradient without any error."""
        dist = torch.distributed
        for t in (self.vae.theta.data, self.vae.bn_state, self.m, self.v):
            dist.broadcast(t, src=src, group=self.pg)
        meta = torch.tensor([self.step_count, self.vae.num_batches_tracked], dtype=torch.int64, device=self.m.device)
        dist.broadcast(meta, src=src, group=self.pg)
        self.step_count, self.vae.num_batches_tracked = int(meta[0].item()), int(meta[1].item())

    def step(self, x, pred, eps):
        """x (B,3,w,w), pred (B,1), eps (B,32): contiguous fp32 device tensors."""
        v, h, B = self.vae, self.h, x.shape[0]
        theta = v.theta.data
        v._stamp_workspace()               # an autograd graph of the same VAE still pending is now stale (nets.py)
        h.forward(B, x, pred, eps, theta, v.bn_state, self.mu, self.logvar, self.recon, self.ws, train=True)
        h.loss(B, x, self.mu, self.logvar, self.recon, self.ws, self.scalars, self.d_recon, self.d_mu, self.d_logvar)
        if self.world_size > 1 and self.overlap:
            # bucketed all-reduce overlapped with backward (torch DDP's scheme on the flat buffer): the
            # three buckets are contiguous ranges, each reduced (sum, RCCL) while the next phase computes
            works = []
            for ph in range(3):
                h.backward_phase(ph, B, x, pred, eps, theta, self.logvar, self.recon, self.d_recon, self.d_mu,
                                 self.d_logvar, self.ws, self.grads)
                off, n = self.buckets[ph]
                if self.grads16 is not None:
                    h.grads_to_bf16(self.grads[off:off + n], self.grads16[off:off + n])
                    works.append(torch.distributed.all_reduce(self.grads16[off:off + n], group=self.pg, async_op=True))
                else:
                    works.append(torch.distributed.all_reduce(self.grads[off:off + n], group=self.pg, async_op=True))
            self._exposed_begin()
            for wk in works:
                wk.wait()                     # nccl: the compute stream waits for the collective's stream
            self._exposed_end()
            if self.grads16 is not None:
                h.grads_from_bf16(self.grads16, self.grads)
        else:
            h.backward(B, x, pred, eps, theta, self.logvar, self.recon, self.d_recon, self.d_mu, self.d_logvar,
                       self.ws, self.grads)
            if self.world_size > 1:
                self._exposed_begin()
                if self.grads16 is not None:
                    h.grads_to_bf16(self.grads, self.grads16)
                    torch.distributed.all_reduce(self.grads16, group=self.pg)
                    h.grads_from_bf16(self.grads16, self.grads)
                else:
                    torch.distributed.all_reduce(self.grads, group=self.pg)      # one flat RCCL all-reduce (sum)
                self._exposed_end()
        self.step_count += 1
        v.num_batches_tracked += 1
        h.adam_step(theta, self.grads, self.m, self.v, self.step_count, self.lr, self.betas[0], self.betas[1],
                    self.eps, grad_scale=1.0 / self.world_size)
        return self.scalars


    def fit_u8(self, frames_u8, critic, batch_size, epochs=1, generator=None, shuffle=True):
        """The `-train` loop (vae.py:40-58) over a host uint8 dataset (N,w,w,3), fused step + overlapped feeder:
        batch i+1 is gathered, copied (pinned, side stream) while step i computes; uint8 -> fp32 CHW/255 and the
        critic run on the GPU.  eps ~ N(0,1) from `generator` (device).  Returns the loss scalars of the last step."""
        from .feeder import FrameFeeder
        dev = self.vae.theta.device
        feeder = FrameFeeder(frames_u8, batch_size, dev, self.h, critic=critic)
        n, scal = frames_u8.shape[0], None
        for _ in range(epochs):
            idx = np.arange(n)
            if shuffle:
                np.random.shuffle(idx)
            for images, preds in feeder.batches([idx[b:b + batch_size] for b in range(0, n, batch_size)]):
                eps = torch.randn(images.shape[0], P.latent_dim, device=dev, generator=generator)
                scal = self.step(images, preds, eps)
        return scal

    # time between "backward is done" and "the reduced gradient is usable" on the compute stream = the part
    # of the all-reduce that backward did not hide (bench.py: allreduce_exposed_us)
    def _exposed_begin(self):
        if self.measure_exposed:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()

    def _exposed_end(self):
        if self.measure_exposed:
            self._ev[1].record()
            self.exposed_ms.append(self._ev)
            if len(self.exposed_ms) > 1024:          # ring: a long run without exposed_us() keeps the newest samples only
                del self.exposed_ms[:512]

    def exposed_us(self):
        """Mean exposed all-reduce time per step (µs) over the steps taken with measure_exposed; syncs."""
        if not self.exposed_ms:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.exposed_ms]
        self.exposed_ms = []
        return 1e3 * sum(ms) / len(ms)


def synthetic_dataset(n_frames, width=P.w, seed=1234):
    x, _, _ = synth.make_batch(seed, 0, n_frames, width)
    return [x[i:i + 1] for i in range(n_frames)]


ENCODER_FILE, DECODER_FILE = "vae_encoder.pt", "vae_decoder.pt"      # vae_parameters.py:25-26 (under saved-networks/)


def save_networks(vae, directory):
    """End of `-train` (vae.py:162-163): torch.save(vae.encoder.state_dict(), ENCODER_PATH) and the same for the decoder —
    the reference's key names and layouts (OIHW conv weights, (C,H,W)-ordered fc columns / decoder_input rows, BatchNorm
    running statistics), so the two files load into the reference's own modules with strict=True and into
    `VariationalAutoencoder.encoder / .decoder.load_state_dict` here.  Returns the two paths."""
    os.makedirs(directory, exist_ok=True)
    enc, dec = os.path.join(directory, ENCODER_FILE), os.path.join(directory, DECODER_FILE)
    torch.save({k: v.detach().cpu() for k, v in vae.encoder.state_dict().items()}, enc)
    torch.save({k: v.detach().cpu() for k, v in vae.decoder.state_dict().items()}, dec)
    return enc, dec


def load_networks(vae, directory, device=None):
    """load_vae_network (vae_utility.py:345-361) for the two files save_networks / the reference wrote."""
    vae.encoder.load_state_dict(torch.load(os.path.join(directory, ENCODER_FILE), map_location=device or "cpu"))
    vae.decoder.load_state_dict(torch.load(os.path.join(directory, DECODER_FILE), map_location=device or "cpu"))
    return vae


def main(argv=None):
    ap = argparse.ArgumentParser(description="Critic-VAE -train on synthetic frames (vae.py:154-163)")
    ap.add_argument("--save", metavar="DIR", default=None,
                    help="write DIR/vae_encoder.pt and DIR/vae_decoder.pt when training ends (vae.py:162-163; the reference's "
                         "DIR is saved-networks/)")
    ap.add_argument("-train", action="store_true")
    ap.add_argument("--synthetic", type=int, default=1024, help="number of synthetic frames")
    ap.add_argument("--batch", type=int, default=P.batch_size)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--critic", default="random", help="'random' scalars (BASELINE config 1), 'synth' = the HIP "
                    "critic with generator weights, or a path to a reference critic checkpoint (.pt)")
    args = ap.parse_args(argv)
    if not args.train:
        ap.error("only -train is implemented (the hot path); see SURVEY.md §8 for scope")
    if not torch.cuda.is_available():
        raise SystemExit("critic-vae_amd needs an MI355X: the HIP library has no CPU fallback")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    device = torch.device("cuda:0")
    vae = VariationalAutoencoder(max_batch=args.batch, seed=args.seed).to(device)
    dset = synthetic_dataset(args.synthetic)
    if args.critic == "random":
        critic_fn = lambda im: torch.rand(im.shape[0], 1, device=im.device)      # noqa: E731
    else:                                  # critic.evaluate(images), vae.py:50 / vae_utility.py:363-370
        from .critic import Critic
        critic = Critic(handle=vae.handle).to(device)
        sd = {k: torch.from_numpy(v) for k, v in synth.make_critic_params(args.seed).items()} \
            if args.critic == "synth" else torch.load(args.critic, map_location="cpu")
        critic.load_state_dict(sd)
        critic_fn = critic.evaluate
    t0 = time.time()
    _, hist = train(vae, dset, critic_fn, device,
                    epochs=args.epochs, batch_size=args.batch, log_n=args.batch * 8)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{args.epochs * args.synthetic / dt:.1f} images/s over {args.epochs} epoch(s)")
    if args.save:
        enc, dec = save_networks(vae, args.save)
        print(f"saved {enc} and {dec}")
    return hist


if __name__ == "__main__":
    main()
