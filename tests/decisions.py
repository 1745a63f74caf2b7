"""Discrete decisions of a training step (max-pool argmax, ReLU masks) — read back from the HIP path's saved
tensors and from the oracle's taps — and the parity check built on them.

Two correct fp32 implementations agree to round-off everywhere, but a max-pool window whose two largest
values differ by an ulp, or a ReLU input within an ulp of zero, may be decided differently; the gradient
then moves by the full local gradient at that unit.  Instead of loosening the relative bound for that,
the tests (1) count the flipped units and show each one sits inside fp32 round-off of a tie, and
(2) compare every gradient element against the oracle run with the HIP path's decisions imposed
(oracle.train_step(decisions=...)), at SURVEY.md A.5's 1e-4 of the tensor's max.
"""
import torch
import torch.nn.functional as F

from critic_vae_amd import synth
from oracle import cvae_oracle as orc

ENC = [(32, 64), (64, 32), (128, 16), (256, 8)]       # (channels, conv output size at width 64)
DEC = [(128, 4), (64, 8), (32, 16), (32, 32)]
# |difference to a tie| of a flipped unit, in units of the normalised / pre-activation value (O(1)), per precision mode.  Why a bound and where
# it comes from: if the HIP step picks window position a and the oracle position b, then n_hip(a) >= n_hip(b) and n_orc(b) >= n_orc(a), hence
# 0 <= n_orc(b) - n_orc(a) <= 2 E with E = max |n_hip - n_orc| over the layer — a flip further from a tie than twice the two sides' largest
# value deviation cannot be round-off (the same for a ReLU input: |n_orc| <= E).  value_deviation() measures E per layer in the run under test
# and check_step_against_oracle asserts gap <= 2 E for every flip; TIE_TOL bounds E itself from the measured distribution over data seeds
# 1234 / 7 / 11 / 21 / 33 at 64x64 (B = 32) and 128x128 (B = 3), profiles/r05_tie_gap_study.txt (tests/../profiles/experiments/tie_gap_study.py):
# worst flipped-unit gap x 2, rounded up (measured worst: f32 6.6e-6, bf16x9 / bf16x6 1.0e-5; E itself is 1-2e-5 in the encoder blocks in ALL
# three modes — BatchNorm's 1/std amplifies the convs' summation-order noise alike — and every observed flip sat within 0.35 of 2 E).  The
# fp32-MFMA mode sums in the reference's k order; the emulation modes add the partial products of a 16-channel block smallest first (bf16x6
# drops three of nine), a different but equally round-off-sized ordering.
TIE_TOL_BY_MODE = {"f32": 2e-5, "bf16x9": 2e-5, "bf16x6": 2e-5}
TIE_TOL = TIE_TOL_BY_MODE["f32"]


def tie_tol(vae):
    return TIE_TOL_BY_MODE.get(getattr(vae.handle, "precision", "f32"), TIE_TOL)


def value_deviation(vae, B, taps):
    """E_l = max |n_hip - n_orc| of the normalised pre-pool values of encoder block l (what the pool / ReLU decisions are taken on), and of the
    decoder's pre-activations recomputed from... the stored o_i (post-ReLU: compared where both sides are positive)."""
    ws, h, k = vae._workspace(B), vae.handle, vae.width // 64
    out = {}
    for l, (c, s) in enumerate(ENC):
        s *= k
        y = h.ws_view(ws, B, f"y{l}", B * s * s * c).view(B, s, s, c).permute(0, 3, 1, 2).double()
        coef = h.ws_view(ws, B, f"coef{l}", c * 4).view(c, 4).double()
        n = (y * coef[:, 0].view(1, c, 1, 1) + coef[:, 1].view(1, c, 1, 1)).float().cpu()
        out[f"enc{l}"] = float((n - taps[f"enc_n{l}"].detach()).abs().max())
    for i, (c, s) in enumerate(DEC):
        o = h.ws_view(ws, B, f"o{i}", B * s * k * s * k * c).view(B, s * k, s * k, c).permute(0, 3, 1, 2).cpu()
        pre = taps[f"dec_pre{i}"].detach()
        both = (o > 0) & (pre > 0)
        out[f"dec{i}"] = float((o - pre)[both].abs().max()) if bool(both.any()) else 0.0
    return out


def hip_decisions(vae, B):
    """Max-pool / ReLU decisions the HIP step actually took, read from what its kernels WROTE: the pooled output a_l
    is matched against the four candidates of its window (the first candidate whose value the kernel stored is the
    window's first maximum), so a kernel whose normalisation expression differed from the candidates recomputed here
    would leave windows without any matching candidate — asserted below — instead of silently imposing decisions the
    kernel never made."""
    ws, h, k = vae._workspace(B), vae.handle, vae.width // 64

    def view(name, c, s):
        return h.ws_view(ws, B, name, B * s * s * c).view(B, s, s, c).permute(0, 3, 1, 2)

    dec = {}
    for l, (c, s) in enumerate(ENC):
        s *= k
        y = view(f"y{l}", c, s).double()
        a = view(f"a{l}", c, s // 2)
        coef = h.ws_view(ws, B, f"coef{l}", c * 4).view(c, 4).double()
        # candidates: fmaf(y, scale, shift), i.e. the correctly rounded fp32 value of y*scale + shift
        n = (y * coef[:, 0].view(1, c, 1, 1) + coef[:, 1].view(1, c, 1, 1)).float()
        cand = torch.stack([n[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], dim=-1)       # scan order
        host_first = cand.argmax(dim=-1)                       # torch.argmax: index of the first maximal value
        if l < 3:                                              # ReLU blocks: a = max(max_p n_p, 0), stored exactly
            au = a.unsqueeze(-1)
            ulp = (torch.nextafter(au.abs(), torch.full_like(au, float("inf"))) - au.abs())
            # exact match preferred; one ulp allowed because the candidates here are double products rounded twice
            # (to double, then to float), which differs from the kernel's single-rounding fmaf once in ~2^30 values
            hit = 2.0 * (cand == au).float() + ((cand - au).abs() <= ulp).float()
            live = a > 0                                       # a == 0: the gradient is masked, the decision is moot
            assert bool((hit.max(-1).values > 0)[live].all()), f"block {l}: a pooled output equals none of its window's candidates"
            pos = torch.where(live, hit.argmax(dim=-1), host_first)
            top = cand.max(-1).values
            assert bool((cand.gather(-1, pos.unsqueeze(-1)).squeeze(-1) >= top - ulp.squeeze(-1))[live].all()), \
                f"block {l}: the stored value is not the window maximum"
        else:                                                  # Tanh block: a = tanh(max); tanh is monotone, compare in value
            pos = host_first
            err = (torch.tanh(cand.max(-1).values.double()) - a.double()).abs().max().item()
            assert err < 2e-6, f"block 3: a3 is not tanh(max of the window candidates): {err:.2e}"
        wo = s // 2
        oy = torch.arange(wo, device=pos.device).view(1, 1, wo, 1)
        ox = torch.arange(wo, device=pos.device).view(1, 1, 1, wo)
        dec[f"pool{l}"] = ((2 * oy + pos // 2) * s + 2 * ox + pos % 2).cpu()     # flat index in the (s x s) plane, as max_pool2d returns
        dec[f"relu_enc{l}"] = (a > 0).cpu()
    for i, (c, s) in enumerate(DEC):
        dec[f"relu_dec{i}"] = (view(f"o{i}", c, s * k) > 0).cpu()
    return dec


def oracle_decisions(taps):
    dec = {}
    for l in range(4):
        dec[f"pool{l}"] = F.max_pool2d(taps[f"enc_n{l}"].detach(), 2, return_indices=True)[1]
        dec[f"relu_enc{l}"] = taps[f"enc_a{l}"].detach() > 0
    for i in range(4):
        dec[f"relu_dec{i}"] = taps[f"dec_o{i}"].detach() > 0
    return dec


def flips(d_hip, d_orc, taps):
    """[(decision name, number of flipped units, largest distance to a tie among them)]"""
    out = []
    for k in d_hip:
        f = d_hip[k] != d_orc[k]
        nf = int(f.sum())
        if not nf:
            continue
        idx = int(k[-1])
        if k.startswith("pool"):
            n = taps[f"enc_n{idx}"].detach()
            a = n.flatten(2).gather(2, d_hip[k].flatten(2)).view(f.shape)
            b = n.flatten(2).gather(2, d_orc[k].flatten(2)).view(f.shape)
            gap = float((a - b).abs()[f].max())
        elif k.startswith("relu_enc"):
            gap = float(F.max_pool2d(taps[f"enc_n{idx}"].detach(), 2).abs()[f].max())
        else:
            gap = float(taps[f"dec_pre{idx}"].detach().abs()[f].max())
        out.append((k, nf, gap))
    return out


def is_pre_bn_bias(name):
    """Conv biases followed by train-mode BatchNorm: true gradient 0, reference value pure round-off."""
    return name.startswith("encoder.model.") and name.endswith(".bias") and int(name.split(".")[2]) % 4 == 0


def check_step_against_oracle(vae, x, pred, eps, B, wseed=0, tol=1e-4, rel=1e-4, max_flips=64, verbose=False):
    """After vae's forward+loss+backward on (x, pred, eps): outputs / loss / every gradient element at `tol`
    absolute vs the plain oracle; flipped decisions counted and shown to be ties; every gradient element at
    `rel` x the tensor's max vs the oracle with the HIP decisions imposed.  Returns a small report."""
    W = vae.width
    params = synth.make_params(wseed, W)
    p = orc.to_torch(params, requires_grad=True)
    taps = {}
    o = orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p), taps=taps)
    g_hip = {k: v.detach().cpu().double() for k, v in vae.reference_grads().items()}
    rep = {"abs": 0.0, "rel_plain": 0.0, "rel_forced": 0.0, "flips": []}
    if not torch.isfinite(o["total_loss"]):
        return None, o
    for k, v in p.items():
        e = float((g_hip[k] - v.grad.double()).abs().max())
        rep["abs"] = max(rep["abs"], e)
        assert e <= tol, f"{k}: abs err {e:.3e}"
        if not is_pre_bn_bias(k):
            rep["rel_plain"] = max(rep["rel_plain"], e / max(float(v.grad.abs().max()), 1e-30))
    d_hip, d_orc = hip_decisions(vae, B), oracle_decisions(taps)
    rep["flips"] = flips(d_hip, d_orc, taps)
    n_units = sum(v.numel() for v in d_hip.values())
    n_flips = sum(f[1] for f in rep["flips"])
    assert n_flips <= max_flips, f"{n_flips} decision flips among {n_units} units: {rep['flips']}"
    rep["dev"] = value_deviation(vae, B, taps)
    for k, nf, gap in rep["flips"]:
        if not k.startswith("relu_dec"):               # encoder decisions: E is exact there (decoder: measured on the units both sides keep, a proxy)
            e = rep["dev"]["enc" + k[-1]]
            assert gap <= 2.0 * e + 1e-12, f"{k}: {nf} flipped unit(s) up to {gap:.3e} away from a tie, but the two sides' values differ by at most {e:.3e}"
        assert gap <= tie_tol(vae), f"{k}: {nf} flipped unit(s) up to {gap:.3e} away from a tie — not round-off ({getattr(vae.handle, 'precision', 'f32')} bound {tie_tol(vae):.0e})"
    if n_flips:
        p2 = orc.to_torch(params, requires_grad=True)
        orc.train_step(p2, x, pred, eps, bn_state=orc.new_bn_state(p2), decisions=d_hip)
        want = {k: v.grad.double() for k, v in p2.items()}
    else:
        want = {k: v.grad.double() for k, v in p.items()}
    for k, w in want.items():
        if is_pre_bn_bias(k):
            continue
        scale = max(float(w.abs().max()), 1e-30)
        e = float((g_hip[k] - w).abs().max())
        rep["rel_forced"] = max(rep["rel_forced"], e / scale)
        assert e <= rel * scale, f"{k}: err {e:.3e} vs max|g| {scale:.3e} (decisions imposed; {n_flips} flips)"
    if verbose:
        print(f"B={B} W={W}: abs {rep['abs']:.2e}  rel(plain) {rep['rel_plain']:.2e}  flips {rep['flips']}  "
              f"rel(decisions imposed) {rep['rel_forced']:.2e}")
    return rep, o
