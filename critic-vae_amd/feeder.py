"""Overlapped frame feeder for the `-train` loop (vae.py:44-50 + vae_utility.py:324-343).

The reference gathers a batch from the host dataset, converts it to fp32 CHW/255 on the CPU, copies
49 152 B per frame over PCIe synchronously from pageable memory and runs the critic — all on the
critical path of the step.  Here the host keeps the frames as the environment delivers them, uint8
HWC (12 288 B per frame: 4x fewer PCIe bytes), and each batch travels

    host gather (numpy take into a PINNED staging buffer)  ->  non-blocking H2D on a SIDE stream
    ->  event  ->  compute stream: cvae_preprocess_u8 (uint8 HWC -> fp32 CHW/255) -> cvae_critic_forward

with two staging / device buffer sets: batch i+1 is gathered and copied right after the caller has
enqueued step i (its launches are asynchronous), so both run under step i's kernels.
A buffer set is re-used only after the compute stream has consumed it (event hand-off both ways).
"""
import numpy as np
import torch

from .critic import Critic


class FrameFeeder:
    """Iterate `(images, preds)` device batches over a host uint8 dataset.

    frames_u8 : np.ndarray (N, W, W, 3) uint8 on the host (what preprocess_observation receives).
    handle    : critic_vae_amd.lib.Handle whose cvae_preprocess_u8 / cvae_critic_forward are used (the VAE's).
    critic    : critic_vae_amd.critic.Critic (HIP critic; `preds = critic.evaluate(images)`, vae.py:50),
                or a callable images -> (B,1), or None (preds = zeros).
    order     : iterable of index arrays, one per batch (the loop's shuffled epoch_indices slices).
    depth     : staging sets in flight (2 = double buffer).
    """

    def __init__(self, frames_u8, batch_size, device, handle, critic=None, depth=2):
        if not (isinstance(frames_u8, np.ndarray) and frames_u8.dtype == np.uint8 and frames_u8.ndim == 4 and frames_u8.shape[3] == 3):
            raise ValueError("frames_u8 must be a host uint8 array of shape (N, W, W, 3)")
        # cvae_preprocess_u8 / cvae_critic_forward index with the HANDLE's width and trust the caller's batch: a mismatch
        # here would be an out-of-bounds access on the GPU, not a Python error
        if not (frames_u8.shape[1] == frames_u8.shape[2] == handle.width):
            raise ValueError(f"frames are {frames_u8.shape[1]}x{frames_u8.shape[2]}, the handle was created for "
                             f"{handle.width}x{handle.width} frames")
        if not 1 <= int(batch_size) <= handle.max_batch:
            raise ValueError(f"batch_size {batch_size} outside 1..max_batch ({handle.max_batch}) of the handle")
        self.frames = frames_u8
        self.B, self.W = int(batch_size), int(frames_u8.shape[1])
        self.device = torch.device(device)
        self.critic = critic
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets = []
        for _ in range(depth):
            pin = torch.empty((self.B, self.W, self.W, 3), dtype=torch.uint8).pin_memory()
            self.sets.append({
                "pin": pin, "pin_np": pin.numpy(),
                "dev_u8": torch.empty((self.B, self.W, self.W, 3), dtype=torch.uint8, device=self.device),
                "x": torch.empty((self.B, 3, self.W, self.W), device=self.device),
                "pred": torch.zeros((self.B, 1), device=self.device),
                "copied": torch.cuda.Event(), "consumed": torch.cuda.Event(), "n": 0, "used": False})
        self._handle = handle

    # -- stage 1 (host + side stream): gather into pinned memory, async H2D
    def _stage(self, s, idx):
        n = len(idx)
        if s["used"]:
            s["consumed"].synchronize()          # the previous occupant of this set has been pre-processed
        np.take(self.frames, idx, axis=0, out=s["pin_np"][:n])
        with torch.cuda.stream(self.copy_stream):
            s["dev_u8"][:n].copy_(s["pin"][:n], non_blocking=True)
            s["copied"].record(self.copy_stream)
        s["n"], s["used"] = n, True

    # -- stage 2 (compute stream): uint8 HWC -> fp32 CHW/255, critic
    def _finish(self, s):
        n = s["n"]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(s["copied"])
        x = s["x"][:n]
        self._handle.preprocess_u8(n, s["dev_u8"][:n], x)
        s["consumed"].record(cur)
        if self.critic is None:
            pred = s["pred"][:n]
        elif isinstance(self.critic, Critic):
            pred = s["pred"][:n]
            self._handle.critic_forward(n, x, self.critic.flat, pred)
        else:
            pred = self.critic(x)
        return x, pred

    def batches(self, order):
        """Yield (images fp32 (n,3,W,W), preds (n,1)) for every index array of `order`.  Batch i+1 is gathered
        and copied right AFTER batch i has been handed out, i.e. once the caller has enqueued step i: the host
        gather and the PCIe copy then run under step i's kernels."""
        order = list(order)
        if not order:
            return
        k = len(self.sets)
        self._stage(self.sets[0], order[0])
        for i in range(len(order)):
            out = self._finish(self.sets[i % k])
            yield out
            if i + 1 < len(order):
                self._stage(self.sets[(i + 1) % k], order[i + 1])
