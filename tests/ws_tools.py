"""Helpers for tests that look at tensors inside a handle's workspace."""
import torch


def recompute_d_y0(h, ws, B, width=64, bf16_storage=False):
    """d_y0 is never materialised by the step: E1's weight-gradient kernel applies block 0's BatchNorm / pool / ReLU
    backward while it stages its tiles (conv_thin.hip, E1Fuse).  Write it into its workspace slot with the stand-alone
    BatchNorm-backward op from the y0 / a0 / d_a0 / coef0 the step left there, so that tests can compare it.
    bf16_storage: the handle keeps activations as bf16 (two elements per workspace float)."""
    per = 2 if bf16_storage else 1

    def sl(name, n_elems, per_float=per):
        off = h.lib.cvae_ws_offset(h.h, B, name.encode())
        assert off >= 0, name
        return ws[off:off + n_elems // per_float]

    n_full, n_pool = B * width * width * 32, B * (width // 2) * (width // 2) * 32
    junk = torch.empty(3 * 32, device=ws.device)
    h.op_bn_pool_act_bwd(0, B, sl("y0", n_full), sl("a0", n_pool), sl("d_a0", n_pool), sl("coef0", 128, 1),
                         junk[:32], sl("d_y0", n_full), junk[32:64], junk[64:], None,
                         torch.empty(h.op_scratch_floats(B), device=ws.device))
    torch.cuda.synchronize()
