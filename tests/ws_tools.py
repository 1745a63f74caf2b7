"""Helpers for tests that look at tensors inside a handle's workspace."""
import torch


def recompute_y0(h, ws, B, x, theta, width=64):
    """bf16 mode does not store y0 either (round 3): the forward's second E1 pass writes only the pooled a0, and E1's
    weight-gradient kernel runs the 75-tap conv again on the tiles it stages.  Write y0 into its workspace slot with the
    stand-alone conv op (layer 0 follows the handle's storage type: bf16 here) from the frames and the E1 parameters."""
    off = h.lib.cvae_ws_offset(h.h, B, b"y0")
    assert off >= 0
    ow, nw = h.layout["enc0.w"]
    ob, nb = h.layout["enc0.b"]
    part = torch.empty(h.op_bn_partial_floats(0, B), device=ws.device)
    h.op_conv_fwd(0, B, x, theta[ow:ow + nw], theta[ob:ob + nb], ws[off:off + B * width * width * 32 // 2], part)
    torch.cuda.synchronize()


def recompute_d_y0(h, ws, B, width=64, bf16_storage=False, x=None, theta=None):
    """d_y0 is never materialised by the step: E1's weight-gradient kernel applies block 0's BatchNorm / pool / ReLU
    backward while it stages its tiles (conv_thin.hip, E1Fuse).  Write it into its workspace slot with the stand-alone
    BatchNorm-backward op from the y0 / a0 / d_a0 / coef0 the step left there, so that tests can compare it.  The
    workspace has no d_y0 slot in the default (fused) configuration: the result is RETURNED as a flat buffer in the handle's
    storage type (fp32 elements, or bf16 elements packed two per float).
    bf16_storage: the handle keeps activations as bf16 (two elements per workspace float); y0 is then recomputed first
    (needs x and theta)."""
    per = 2 if bf16_storage else 1
    if bf16_storage:
        recompute_y0(h, ws, B, x, theta, width)

    def sl(name, n_elems, per_float=per):
        off = h.lib.cvae_ws_offset(h.h, B, name.encode())
        assert off >= 0, name
        return ws[off:off + n_elems // per_float]

    n_full, n_pool = B * width * width * 32, B * (width // 2) * (width // 2) * 32
    junk = torch.empty(3 * 32, device=ws.device)
    d_y0 = torch.empty(n_full // per, device=ws.device)
    h.op_bn_pool_act_bwd(0, B, sl("y0", n_full), sl("a0", n_pool), sl("d_a0", n_pool), sl("coef0", 128, 1),
                         junk[:32], d_y0, junk[32:64], junk[64:], None,
                         torch.empty(h.op_scratch_floats(B), device=ws.device))
    torch.cuda.synchronize()
    return d_y0
