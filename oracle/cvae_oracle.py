"""ORACLE — CPU restatement of the Critic-VAE training step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (critic-vae_amd/) never does and fails loudly without its HIP
library.

What it restates (reference = /root/reference, PyTorch ATen on CPU, fp32):
  * VariationalEncoder.forward            vae_nets.py:64-111
  * VariationalAutoencoder.reparametrize  vae_nets.py:48-51  (eps is an explicit input)
  * Decoder.forward (evalu=False)         vae_nets.py:113-147
  * MSSIM.{gaussian_window,create_window,ssim,forward}  vae_nets.py:150-247
  * VariationalAutoencoder.vae_loss       vae_nets.py:53-62
  * loss.backward()                       vae.py:57 (torch autograd over the ops above)
  * Adam.step()                           vae.py:36,58 (torch.optim.Adam defaults)
  * Critic.evaluate (eval mode)           critic_net.py:5-69, and preprocess_observation vae_utility.py:324-343

It is written as table-driven functional code over a flat {name: tensor} parameter dict (names
= the reference's state_dict keys, prefixed encoder./decoder.), not as nn.Modules.

Pinning: the reference has no tests/fixtures for this path, so parity is pinned
by outputs of the reference itself run in the build container: tests/golden/make_golden.py
imports /root/reference/vae_nets.py, checks this module against it on identical
weights/inputs/noise, and writes tests/golden/*.npz; tests/test_oracle.py re-checks this module
against those fixtures anywhere (no reference needed).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

K, PAD = 5, 2                       # vae_parameters.py:12-13
LATENT = 32                         # vae_parameters.py:16
KLD_WEIGHT = 0.001                  # vae_parameters.py:17
BN_EPS, BN_MOMENTUM = 1e-5, 0.1     # nn.BatchNorm2d defaults, vae_nets.py:70
MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)   # vae_nets.py:219
MS_WINDOW, MS_SIGMA = 11, 1.5       # vae_nets.py:153, 175
C1, C2 = 0.01 ** 2, 0.03 ** 2       # vae_nets.py:201-203 (img_range fixed to 1.0)

# encoder blocks: (conv state_dict index, bn index, activation)      vae_nets.py:68-88
ENC_BLOCKS = ((0, 1, "relu"), (4, 5, "relu"), (8, 9, "relu"), (12, 13, "tanh"))
# decoder convs: (state_dict index, activation, upsample-after)       vae_nets.py:116-135
DEC_BLOCKS = ((0, "relu", True), (3, "relu", True), (6, "relu", True), (9, "relu", True),
              (12, "tanh", False))


def to_torch(params_np, requires_grad=False):
    out = {}
    for k, v in params_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def new_bn_state(params):
    """running_mean=0, running_var=1, num_batches_tracked=0 per BatchNorm2d (torch defaults)."""
    st = {}
    for _, bi, _ in ENC_BLOCKS:
        c = params[f"encoder.model.{bi}.weight"].shape[0]
        st[f"encoder.model.{bi}.running_mean"] = torch.zeros(c)
        st[f"encoder.model.{bi}.running_var"] = torch.ones(c)
        st[f"encoder.model.{bi}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return st


def _act(x, kind):
    return torch.relu(x) if kind == "relu" else torch.tanh(x)


def encoder(params, x, bn_state=None, train=True, taps=None, decisions=None):
    """vae_nets.py:101-111.  Conv -> BatchNorm(batch stats when train) -> MaxPool2 -> act, x4;
    flatten in (C,H,W) order; fc_mu / fc_var.  `taps` (dict) collects intermediates.

    `decisions` (tests only; default None = the reference's own behaviour): impose the discrete choices of
    another implementation — `pool{l}`: max-pool argmax as flat H*W indices (B,C,H/2,W/2), `relu_enc{l}`: 0/1
    mask of the units ReLU lets through — so that two runs that agree to fp32 round-off everywhere can be
    compared beyond the handful of units whose max / sign decision sits inside that round-off."""
    h = x
    for bi_, (ci, bi, act) in enumerate(ENC_BLOCKS):
        y = F.conv2d(h, params[f"encoder.model.{ci}.weight"], params[f"encoder.model.{ci}.bias"],
                     stride=1, padding=PAD)
        rm = rv = None
        if bn_state is not None:
            rm = bn_state[f"encoder.model.{bi}.running_mean"]
            rv = bn_state[f"encoder.model.{bi}.running_var"]
            if train:
                bn_state[f"encoder.model.{bi}.num_batches_tracked"] += 1
        if not train and bn_state is None:
            raise ValueError("eval-mode BatchNorm needs running statistics")
        n = F.batch_norm(y, rm, rv, params[f"encoder.model.{bi}.weight"],
                         params[f"encoder.model.{bi}.bias"], training=train,
                         momentum=BN_MOMENTUM, eps=BN_EPS)
        if decisions is None:
            h = _act(F.max_pool2d(n, 2), act)
        else:
            idx = decisions[f"pool{bi_}"]
            pooled = n.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            h = pooled * decisions[f"relu_enc{bi_}"].to(pooled.dtype) if act == "relu" else torch.tanh(pooled)
        if taps is not None:
            taps[f"enc_y{bi_}"] = y
            taps[f"enc_n{bi_}"] = n
            taps[f"enc_a{bi_}"] = h
    flat = torch.flatten(h, start_dim=1)
    mu = F.linear(flat, params["encoder.fc_mu.weight"], params["encoder.fc_mu.bias"])
    logvar = F.linear(flat, params["encoder.fc_var.weight"], params["encoder.fc_var.bias"])
    return mu, logvar


def reparametrize(mu, logvar, eps):
    """vae_nets.py:48-51 with the noise supplied by the caller."""
    return mu + eps * torch.exp(0.5 * logvar)


def decoder(params, z, pred, taps=None, decisions=None):
    """vae_nets.py:139-147 (training branch): cat -> Linear -> view(-1,256,s,s) -> convs.
    `decisions["relu_dec{i}"]` (tests only): imposed 0/1 ReLU masks, see encoder()."""
    hcat = torch.cat((z, pred), dim=1)
    h = F.linear(hcat, params["decoder.decoder_input.weight"], params["decoder.decoder_input.bias"])
    side = int(round(math.sqrt(h.shape[1] // 256)))
    h = h.view(-1, 256, side, side)
    if taps is not None:
        taps["dec_h"] = h
    for i, (ci, act, up) in enumerate(DEC_BLOCKS):
        pre = F.conv2d(h, params[f"decoder.model.{ci}.weight"], params[f"decoder.model.{ci}.bias"], stride=1, padding=PAD)
        if decisions is not None and act == "relu":
            h = pre * decisions[f"relu_dec{i}"].to(pre.dtype)
        else:
            h = _act(pre, act)
        if taps is not None:
            taps[f"dec_pre{i}"] = pre
            taps[f"dec_o{i}"] = h
        if up:
            h = F.interpolate(h, scale_factor=2, mode="nearest")     # nn.Upsample default mode
    return h


def ms_window_1d():
    """vae_nets.py:170-173 — note the POSITIVE exponent (edge-heavy 'anti-Gaussian')."""
    g = torch.tensor([math.exp((i - MS_WINDOW // 2) ** 2 / (2 * MS_SIGMA ** 2))
                      for i in range(MS_WINDOW)])
    return g / g.sum()


def ms_window_2d(channels):
    """vae_nets.py:175-179."""
    g = ms_window_1d().unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channels, 1, MS_WINDOW, MS_WINDOW).contiguous()


def ssim_level(img1, img2, window):
    """vae_nets.py:181-215 with size_average=True: returns (mean ssim_map, mean cs_map)."""
    c = img1.shape[1]
    pad = MS_WINDOW // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=c)
    mu2 = F.conv2d(img2, window, padding=pad, groups=c)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=pad, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=pad, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=pad, groups=c) - mu1_mu2
    v1 = 2.0 * s12 + C2
    v2 = s1 + s2 + C2
    cs = torch.mean(v1 / v2)
    ssim_map = ((2 * mu1_mu2 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)
    return ssim_map.mean(), cs


def msssim(img1, img2):
    """vae_nets.py:217-247.  Returns (1 - output, ssim[5], cs[5])."""
    weights = torch.tensor(MS_WEIGHTS, dtype=torch.float32)
    window = ms_window_2d(img1.shape[1])
    sims, css = [], []
    for _ in range(len(MS_WEIGHTS)):
        s, c = ssim_level(img1, img2, window)
        sims.append(s)
        css.append(c)
        img1 = F.avg_pool2d(img1, (2, 2))
        img2 = F.avg_pool2d(img2, (2, 2))
    sims, css = torch.stack(sims), torch.stack(css)
    pow1 = css ** weights
    pow2 = sims ** weights
    output = torch.prod(pow1[:-1] * pow2[-1])        # the reference's quirk, kept (vae_nets.py:246)
    return 1 - output, sims, css


def vae_loss(x, mu, logvar, recon):
    """vae_nets.py:53-62."""
    recon_loss, sims, css = msssim(recon, x)
    kld = torch.mean(-0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp(), dim=1), dim=0)
    kld = kld * KLD_WEIGHT
    return {"total_loss": recon_loss + kld, "recon_loss": recon_loss.detach(), "KLD": kld.detach(),
            "ssim_levels": sims.detach(), "cs_levels": css.detach()}


def train_step(params, x, pred, eps, bn_state=None, taps=None, decisions=None):
    """forward + loss + backward of vae.py:53-57 on explicit (x, pred, eps).

    `params` must be leaf tensors with requires_grad=True; their .grad is filled (accumulated,
    like autograd does).  Returns dict with mu, logvar, recon and the loss scalars."""
    mu, logvar = encoder(params, x, bn_state, train=True, taps=taps, decisions=decisions)
    z = reparametrize(mu, logvar, eps)
    recon = decoder(params, z, pred, taps=taps, decisions=decisions)
    losses = vae_loss(x, mu, logvar, recon)
    if taps is not None:
        for t in taps.values():
            t.retain_grad()
        recon.retain_grad(); mu.retain_grad(); logvar.retain_grad(); z.retain_grad()
        taps["z"] = z
    losses["total_loss"].backward()
    out = {"mu": mu, "logvar": logvar, "recon": recon}
    out.update(losses)
    return out


def adam_step(params, state, lr=5e-5, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults as used by vae.py:36 (no weight decay, no amsgrad), restated:
    m=b1 m+(1-b1) g; v=b2 v+(1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    with torch.no_grad():
        for k, p in params.items():
            if p.grad is None:
                continue
            m = state.setdefault("m/" + k, torch.zeros_like(p))
            v = state.setdefault("v/" + k, torch.zeros_like(p))
            m.mul_(b1).add_(p.grad, alpha=1 - b1)
            v.mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


def zero_grad(params):
    for p in params.values():
        p.grad = None


# --------------------------------------------------------------------------------------------
# the frozen critic that supplies `preds` (vae.py:50) and the frame pre-processing
# --------------------------------------------------------------------------------------------
CRITIC_CONVS = ((0, 1, True), (3, 1, True), (6, 1, True), (10, 1, True), (14, 0, False))   # (index, pad, pool)


def critic_forward(cp, x):
    """Critic.forward / evaluate in eval mode (critic_net.py:15-41, 43-69): 4 x [Conv3x3(p=1) -> ReLU ->
    MaxPool2] -> Conv4x4 -> ReLU -> Flatten -> Linear -> ReLU -> Linear -> Sigmoid; Dropout is identity.
    `cp`: {reference state_dict key: tensor}."""
    h = x
    for idx, pad, pool in CRITIC_CONVS:
        h = torch.relu(F.conv2d(h, cp[f"features.{idx}.weight"], cp[f"features.{idx}.bias"], stride=1, padding=pad))
        if pool:
            h = F.max_pool2d(h, 2)
    h = torch.flatten(h, 1)
    h = torch.relu(F.linear(h, cp["crit.1.weight"], cp["crit.1.bias"]))
    return torch.sigmoid(F.linear(h, cp["crit.4.weight"], cp["crit.4.bias"]))


def preprocess_frames(u8_hwc):
    """adjust_values + HWC->CHW of preprocess_observation (vae_utility.py:324-343), batched."""
    return (u8_hwc.to(torch.float32) / 255.0).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------------
# single-op references used by the per-kernel parity tests
# --------------------------------------------------------------------------------------------

def conv5x5(x, w, b=None, upsample_input=False):
    """Conv2d(k=5,s=1,p=2) on NCHW; optional nearest-2x upsample of the input first
    (vae_nets.py:119-131: Upsample precedes the next decoder conv)."""
    if upsample_input:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, w, b, stride=1, padding=PAD)


def bn_pool_act(y, gamma, beta, act):
    """BatchNorm2d(train) -> MaxPool2d(2) -> ReLU/Tanh (vae_nets.py:70-72); also returns the
    batch mean and biased variance."""
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    n = F.batch_norm(y, None, None, gamma, beta, training=True, eps=BN_EPS)
    return _act(F.max_pool2d(n, 2), act), mean, var
