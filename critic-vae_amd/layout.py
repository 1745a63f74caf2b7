"""Reference <-> native parameter layouts.

The library keeps all parameters (and gradients) in one flat fp32 buffer in kernel-native
layouts (include/cvae.h).  The reference's checkpoints / state_dicts (vae.py:162-163,
vae_utility.py:345-361) use PyTorch layouts: conv OIHW, Linear [out,in] with the 256*s*s
bottleneck axis in (C,H,W) flatten order (vae_nets.py:105,144).  These two functions are the
only place that knows both; Adam is elementwise so training in native layout is exact.
"""
import torch

from . import params as P

ENC_CONV = (0, 4, 8, 12)        # encoder.model.{i}   conv ; i+1 = BatchNorm (vae_nets.py:68-88)
DEC_CONV = (0, 3, 6, 9, 12)     # decoder.model.{i}   conv            (vae_nets.py:116-135)


def _conv_to_native(w):          # (O, I, 5, 5) -> [25][I][O]
    return w.permute(2, 3, 1, 0).reshape(25, w.shape[1], w.shape[0])


def _conv_to_ref(n, cin, cout):  # [25*I*O] -> (O, I, 5, 5)
    return n.reshape(5, 5, cin, cout).permute(3, 2, 0, 1).contiguous()


def ref_to_native(layout, total, ref, device=None, dtype=torch.float32):
    """ref: {reference state_dict key (prefixed encoder./decoder.): tensor}.  Returns the flat
    native buffer (`total` floats, padding zero)."""
    any_t = next(iter(ref.values()))
    device = device if device is not None else any_t.device
    flat = torch.zeros(total, dtype=dtype, device=device)

    def put(name, t):
        off, n = layout[name]
        assert t.numel() == n, (name, t.shape, n)
        flat[off:off + n] = t.reshape(-1).to(device=device, dtype=dtype)

    for l, ci in enumerate(ENC_CONV):
        put(f"enc{l}.w", _conv_to_native(ref[f"encoder.model.{ci}.weight"]))
        put(f"enc{l}.b", ref[f"encoder.model.{ci}.bias"])
        put(f"enc{l}.gamma", ref[f"encoder.model.{ci + 1}.weight"])
        put(f"enc{l}.beta", ref[f"encoder.model.{ci + 1}.bias"])
    wcat = torch.cat((ref["encoder.fc_mu.weight"], ref["encoder.fc_var.weight"]), 0)   # (64, K)
    K = wcat.shape[1]
    s = int(round((K // 256) ** 0.5))
    put("fc.w", wcat.reshape(64, 256, s, s).permute(2, 3, 1, 0).reshape(K, 64))
    put("fc.b", torch.cat((ref["encoder.fc_mu.bias"], ref["encoder.fc_var.bias"]), 0))
    for i, ci in enumerate(DEC_CONV):
        put(f"dec{i}.w", _conv_to_native(ref[f"decoder.model.{ci}.weight"]))
        put(f"dec{i}.b", ref[f"decoder.model.{ci}.bias"])
    wd = ref["decoder.decoder_input.weight"]                                            # (K, 33)
    put("decin.w", wd.reshape(256, s, s, P.latent_dim + 1).permute(3, 1, 2, 0).reshape(P.latent_dim + 1, K))
    put("decin.b", ref["decoder.decoder_input.bias"].reshape(256, s, s).permute(1, 2, 0).reshape(K))
    return flat


def native_to_ref(layout, flat, dims=P.dims):
    """Inverse of ref_to_native (works for parameters and for gradients)."""
    def get(name):
        off, n = layout[name]
        return flat[off:off + n]

    out = {}
    enc_ch = (P.ch,) + tuple(dims)
    for l, ci in enumerate(ENC_CONV):
        out[f"encoder.model.{ci}.weight"] = _conv_to_ref(get(f"enc{l}.w"), enc_ch[l], enc_ch[l + 1])
        out[f"encoder.model.{ci}.bias"] = get(f"enc{l}.b").clone()
        out[f"encoder.model.{ci + 1}.weight"] = get(f"enc{l}.gamma").clone()
        out[f"encoder.model.{ci + 1}.bias"] = get(f"enc{l}.beta").clone()
    K = layout["fc.w"][1] // 64
    s = int(round((K // 256) ** 0.5))
    wcat = get("fc.w").reshape(s, s, 256, 64).permute(3, 2, 0, 1).reshape(64, K)
    out["encoder.fc_mu.weight"] = wcat[:32].contiguous()
    out["encoder.fc_var.weight"] = wcat[32:].contiguous()
    b = get("fc.b")
    out["encoder.fc_mu.bias"] = b[:32].clone()
    out["encoder.fc_var.bias"] = b[32:].clone()
    dec_ch = (dims[3], dims[2], dims[1], dims[0], dims[0], P.ch)
    for i, ci in enumerate(DEC_CONV):
        out[f"decoder.model.{ci}.weight"] = _conv_to_ref(get(f"dec{i}.w"), dec_ch[i], dec_ch[i + 1])
        out[f"decoder.model.{ci}.bias"] = get(f"dec{i}.b").clone()
    L = P.latent_dim + 1
    out["decoder.decoder_input.weight"] = get("decin.w").reshape(L, s, s, 256).permute(3, 1, 2, 0).reshape(K, L).contiguous()
    out["decoder.decoder_input.bias"] = get("decin.b").reshape(s, s, 256).permute(2, 0, 1).reshape(K).contiguous()
    return out


BN_OFF = (0, 32, 96, 224)
BN_TOTAL = 480


def bn_state_to_ref(bn_state, num_batches_tracked):
    """[mean(480) | var(480)] -> reference BatchNorm buffers."""
    out = {}
    for l, ci in enumerate(ENC_CONV):
        c = P.dims[l]
        out[f"encoder.model.{ci + 1}.running_mean"] = bn_state[BN_OFF[l]:BN_OFF[l] + c].clone()
        out[f"encoder.model.{ci + 1}.running_var"] = bn_state[BN_TOTAL + BN_OFF[l]:BN_TOTAL + BN_OFF[l] + c].clone()
        out[f"encoder.model.{ci + 1}.num_batches_tracked"] = torch.tensor(num_batches_tracked, dtype=torch.long)
    return out
