"""GPU parity: every HIP kernel, through the C-ABI, against the oracle's single-op references.

Tolerance: north_star asks 1e-4 fp32 on outputs; gradients of tensors whose magnitude is far
from 1 are additionally checked relative to the tensor's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from critic_vae_amd import lib as cvlib
from critic_vae_amd import synth
from oracle import cvae_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4
# (cin, cout, h_out, up)  — vae_nets.py:69-84, 117-133
LAYERS = [(3, 32, 64, 0), (32, 64, 32, 0), (64, 128, 16, 0), (128, 256, 8, 0),
          (256, 128, 4, 0), (128, 64, 8, 1), (64, 32, 16, 1), (32, 32, 32, 1), (32, 3, 64, 1)]


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return cvlib.Handle(64, 64)


def dev(t):
    return t.contiguous().cuda()


def nhwc(t):       # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def to_nchw(t, B, h, c):
    return t.view(B, h, h, c).permute(0, 3, 1, 2).cpu()


def wnat(w):       # OIHW -> [25][I][O]
    return w.permute(2, 3, 1, 0).reshape(25, w.shape[1], w.shape[0]).contiguous().cuda()


def wref(n, cin, cout):
    return n.view(5, 5, cin, cout).permute(3, 2, 0, 1).cpu()


def rnd(name, shape, lo=-1.0, hi=1.0, seed=7):
    return torch.from_numpy(synth.uniform(seed, name, shape, lo, hi))


def check(got, want, what, tol=TOL, rel=False):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() if rel else 1.0
    assert np.isfinite(err) and err <= tol * max(scale, 1e-30), f"{what}: max|err|={err:.3e} scale={scale:.3e}"


@pytest.mark.parametrize("layer", range(9))
@pytest.mark.parametrize("B", [3, 8])
def test_conv_fwd(H, layer, B):
    cin, cout, h, up = LAYERS[layer]
    hs = h // 2 if up else h
    x = rnd(f"x{layer}", (B, cin, hs, hs))
    w = rnd(f"w{layer}", (cout, cin, 5, 5), -0.1, 0.1)
    b = rnd(f"b{layer}", (cout,))
    ref = orc.conv5x5(x, w, b, upsample_input=bool(up))
    if 4 <= layer <= 7:
        ref = torch.relu(ref)
    if layer == 8:
        ref = torch.tanh(ref)
    xin = dev(x) if layer == 0 else nhwc(x)
    out = torch.full((B * h * h * cout,), float("nan"), device="cuda")
    part = torch.zeros(max(H.op_bn_partial_floats(min(layer, 3), B), 1), device="cuda") if layer < 4 else None
    H.op_conv_fwd(layer, B, xin, wnat(w), dev(b), out, part, torch.empty(H.op_scratch_floats(B), device="cuda"))
    torch.cuda.synchronize()
    got = out.view(B, cout, h, h).cpu() if layer == 8 else to_nchw(out, B, h, cout)
    check(got, ref, f"conv_fwd L{layer}")


@pytest.mark.parametrize("B", [37, 64])
def test_conv_fwd_d0_image_major_tiles(H, B):
    """D0 (4x4 images) runs on conv4x4_row_kernel: M tiles of 32 IMAGES x two image rows that skip the zero padding, split-K
    over channel chunks.  Batches that end inside a 32-image group (the cases above stay inside the first one)."""
    cin, cout, h, _ = LAYERS[4]
    x = rnd("x4r", (B, cin, h, h))
    w = rnd("w4r", (cout, cin, 5, 5), -0.1, 0.1)
    b = rnd("b4r", (cout,))
    ref = torch.relu(orc.conv5x5(x, w, b, upsample_input=False))
    out = torch.full((B * h * h * cout,), float("nan"), device="cuda")
    H.op_conv_fwd(4, B, nhwc(x), wnat(w), dev(b), out, None, torch.empty(H.op_scratch_floats(B), device="cuda"))
    torch.cuda.synchronize()
    check(to_nchw(out, B, h, cout), ref, "conv_fwd L4 (ragged image groups)")


@pytest.mark.parametrize("layer", range(4))
@pytest.mark.parametrize("B,ties", [(3, False), (8, True), (5, "tiny_gamma")])
def test_bn_pool_act_fwd_bwd(H, layer, B, ties):
    """ties=True: exact max-pool ties + negative gamma.  "tiny_gamma": channels with gamma = 0 / 3e-3 / -5e-3 —
    the ReLU blocks' pooled-tensor shortcut xhat = (a - beta)/gamma is ill-conditioned there and the kernel
    must fall back to y (dgamma of a zero-gamma channel is sum(g*xhat), not 0)."""
    tiny = ties == "tiny_gamma"
    ties = ties is True
    cin, C, h, _ = LAYERS[layer]
    act = "tanh" if layer == 3 else "relu"
    x = rnd(f"bx{layer}", (B, cin, h, h))
    w = rnd(f"bw{layer}", (C, cin, 5, 5), -0.1, 0.1)
    b = rnd(f"bb{layer}", (C,))
    if ties:       # y = x[co % cin] exactly, x on a coarse grid -> many EXACT ties inside pooling windows
        x = torch.round(x * 2) / 2
        w = torch.zeros_like(w)
        w[torch.arange(C), torch.arange(C) % cin, 2, 2] = 1.0
        b = torch.zeros_like(b)
    gamma = rnd(f"g{layer}", (C,), 0.5, 1.5)
    beta = rnd(f"be{layer}", (C,), -0.5, 0.5)
    if ties:
        gamma[::3] *= -1            # negative scale flips which element is the maximum
    if tiny:
        gamma[0], gamma[5], gamma[9] = 0.0, 3e-3, -5e-3
        beta[0], beta[5], beta[9] = 0.3, 0.5, 0.2          # ReLU lets these channels through
    # conv on the GPU provides y and the BatchNorm partials
    y = torch.empty(B * h * h * C, device="cuda")
    part = torch.zeros(H.op_bn_partial_floats(layer, B), device="cuda")
    H.op_conv_fwd(layer, B, dev(x) if layer == 0 else nhwc(x), wnat(w), dev(b), y, part)
    y_ref = to_nchw(y, B, h, C).clone().requires_grad_(True)
    g_ref, b_ref = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a_ref, mean, var = orc.bn_pool_act(y_ref, g_ref, b_ref, act)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    coef = torch.empty(C * 4, device="cuda")
    a = torch.empty(B * (h // 2) ** 2 * C, device="cuda")
    sc = torch.empty(H.op_scratch_floats(B), device="cuda")
    H.op_bn_pool_act_fwd(layer, B, y, part, dev(gamma), dev(beta), rm, rv, coef, a, sc, True)
    torch.cuda.synchronize()
    check(coef.view(C, 4)[:, 2], mean, "bn mean", 1e-5)
    check(1.0 / coef.view(C, 4)[:, 3] ** 2 - 1e-5, var, "bn var", 1e-5, rel=True)
    check(to_nchw(a, B, h // 2, C), a_ref, f"bn_pool_act fwd L{layer}")
    n = B * h * h
    check(rm, 0.1 * mean, "running_mean", 1e-5)
    check(rv, 0.9 + 0.1 * var * n / (n - 1), "running_var", 1e-5)
    # backward
    da = rnd(f"da{layer}", (B, C, h // 2, h // 2))
    a_ref.backward(da)
    dy = torch.empty_like(y)
    dg, db, dbias = (torch.empty(C, device="cuda") for _ in range(3))
    sc = torch.empty(H.op_scratch_floats(B), device="cuda")
    H.op_bn_pool_act_bwd(layer, B, y, a, nhwc(da), coef, dev(gamma), dy, dg, db, dbias, sc)
    torch.cuda.synchronize()
    check(to_nchw(dy, B, h, C), y_ref.grad, f"bn_pool_act bwd dy L{layer}", rel=True)
    check(dg, g_ref.grad, "dgamma", rel=True)
    check(db, b_ref.grad, "dbeta", rel=True)
    # the conv bias feeds a train-mode BatchNorm: its true gradient is 0 and both sides hold pure
    # summation round-off, so compare against the size of the summed terms
    noise = 2e-6 * y_ref.grad.abs().sum(dim=(0, 2, 3)).max().item()
    check(dbias, y_ref.grad.sum(dim=(0, 2, 3)), "dbias (round-off level)", max(noise, 1e-9))


@pytest.mark.parametrize("layer", range(1, 8))
@pytest.mark.parametrize("B", [3, 8])
def test_conv_dgrad(H, layer, B):
    cin, cout, h, up = LAYERS[layer]
    hs = h // 2 if up else h
    w = rnd(f"w{layer}", (cout, cin, 5, 5), -0.1, 0.1)
    dout = rnd(f"do{layer}", (B, cout, h, h))
    pre = rnd(f"pre{layer}", (B, cin, hs, hs)).requires_grad_(True)
    src = torch.relu(pre) if up else pre
    orc.conv5x5(src, w, None, upsample_input=bool(up)).backward(dout)
    din = torch.full((B * hs * hs * cin,), float("nan"), device="cuda")
    sc = torch.empty(H.op_scratch_floats(B), device="cuda")
    H.op_conv_dgrad(layer, B, nhwc(dout), wnat(w), nhwc(src.detach()) if up else None, din, sc)
    torch.cuda.synchronize()
    check(to_nchw(din, B, hs, cin), pre.grad, f"conv_dgrad L{layer}", rel=True)


@pytest.mark.parametrize("layer", range(8))
@pytest.mark.parametrize("B", [3, 8])
def test_conv_wgrad(H, layer, B):
    cin, cout, h, up = LAYERS[layer]
    hs = h // 2 if up else h
    x = rnd(f"x{layer}", (B, cin, hs, hs))
    w = rnd(f"w{layer}", (cout, cin, 5, 5), -0.1, 0.1).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    dout = rnd(f"do{layer}", (B, cout, h, h))
    orc.conv5x5(x, w, b, upsample_input=bool(up)).backward(dout)
    dw = torch.full((25 * cin * cout,), float("nan"), device="cuda")
    dbias = torch.empty(cout, device="cuda") if layer >= 4 else None
    sc = torch.empty(H.op_scratch_floats(B), device="cuda")
    H.op_conv_wgrad(layer, B, dev(x) if layer == 0 else nhwc(x), nhwc(dout), dw, dbias, sc)
    torch.cuda.synchronize()
    check(wref(dw, cin, cout), w.grad, f"conv_wgrad L{layer}", rel=True)
    if dbias is not None:
        check(dbias, b.grad, f"conv dbias L{layer}", rel=True)


@pytest.mark.parametrize("B", [2, 5])
def test_d4_bwd(H, B):
    w = rnd("w8", (3, 32, 5, 5), -0.1, 0.1).requires_grad_(True)
    b = rnd("b8", (3,)).requires_grad_(True)
    pre = rnd("pre8", (B, 32, 32, 32)).requires_grad_(True)
    o3 = torch.relu(pre)
    recon = torch.tanh(orc.conv5x5(o3, w, b, upsample_input=True))
    d_recon = rnd("dr8", (B, 3, 64, 64))
    recon.backward(d_recon)
    dout = torch.empty(B * 3 * 64 * 64, device="cuda")
    d_o3 = torch.full((B * 32 * 32 * 32,), float("nan"), device="cuda")
    dw = torch.empty(2400, device="cuda")
    db = torch.empty(3, device="cuda")
    sc = torch.empty(H.op_scratch_floats(B), device="cuda")
    H.op_d4_bwd(B, nhwc(o3.detach()), dev(d_recon), dev(recon.detach()), wnat(w.detach()), dout, d_o3, dw, db, sc)
    torch.cuda.synchronize()
    check(dout.view(B, 3, 64, 64), d_recon * (1 - recon.detach() ** 2), "d4 tanh bwd", 1e-5)
    check(to_nchw(d_o3, B, 32, 32), pre.grad, "d4 dgrad", rel=True)
    check(wref(dw, 32, 3), w.grad, "d4 wgrad", rel=True)
    check(db, b.grad, "d4 dbias", rel=True)


def _ms_inputs(tag):
    b = torch.from_numpy(synth.uniform(5, f"ms/{tag}/b", (4, 3, 64, 64)))
    if tag == "nan":
        a = 0.5 - b + 0.01 * torch.from_numpy(synth.uniform(5, f"ms/{tag}/a", (4, 3, 64, 64)))
    else:
        a = torch.from_numpy(synth.uniform(5, f"ms/{tag}/a", (4, 3, 64, 64), -1.0 if tag == "neg" else 0.0, 1.0))
    return a.clone(), b


@pytest.mark.parametrize("tag", ["pos", "neg", "nan"])
def test_msssim(H, golden_dir, tag):
    import os
    fx = np.load(os.path.join(golden_dir, "msssim_ops.npz"))
    a, b = _ms_inputs(tag)
    a_ref = a.clone().requires_grad_(True)
    loss, sims, css = orc.msssim(a_ref, b)
    ws = torch.empty(H.op_msssim_ws_floats(4), device="cuda")
    scal = torch.empty(16, device="cuda")
    d_a = torch.empty(4 * 3 * 64 * 64, device="cuda")
    H.op_msssim(4, dev(a), dev(b), ws, scal, d_a)
    torch.cuda.synchronize()
    s = scal.cpu()
    check(s[3:8], sims, "ssim levels", 2e-5)
    check(s[8:13], css, "cs levels", 2e-5)
    check(s[3:8], torch.from_numpy(fx[f"{tag}/ssim"]), "ssim levels vs reference fixture", 2e-5)
    check(s[8:13], torch.from_numpy(fx[f"{tag}/cs"]), "cs levels vs reference fixture", 2e-5)
    if tag == "nan":
        assert torch.isnan(s[1]) and torch.isnan(loss) and np.isnan(fx["nan/loss"])    # NaN propagates, no clamp
        return
    check(s[1], loss, "msssim loss", 2e-5)
    assert abs(s[1].item() - float(fx[f"{tag}/loss"])) < 2e-5
    assert s[2].item() == 0.0 and abs(s[0].item() - s[1].item()) < 1e-7
    loss.backward()
    check(d_a.view(4, 3, 64, 64), a_ref.grad, "msssim grad", rel=True)


@pytest.mark.parametrize("B", [3, 90])          # 90: 270 planes on 256 persistent workgroups (some walk two planes)
def test_msssim_128_wide_streamed_level_against_oracle(B):
    """Round 4: level 0 of 128x128 frames runs on msssim_stream_kernel (the plane streamed through LDS in 16-row bands,
    msssim.hip).  Loss, the five ssim / cs levels and every element of the gradient against the oracle's MSSIM
    (vae_nets.py:181-247 restated) on frames with structure (a smooth image against a noisy copy: all levels positive)."""
    h = cvlib.Handle(128, B)
    g = torch.Generator().manual_seed(100 + B)
    b = torch.rand(B, 3, 128, 128, generator=g)
    a = (0.7 * b + 0.3 * torch.rand(B, 3, 128, 128, generator=g)).clone()
    a_ref = a.clone().requires_grad_(True)
    loss, sims, css = orc.msssim(a_ref, b)
    assert torch.isfinite(loss)
    ws = torch.empty(h.op_msssim_ws_floats(B), device="cuda")
    scal = torch.empty(16, device="cuda")
    d_a = torch.empty(B * 3 * 128 * 128, device="cuda")
    h.op_msssim(B, dev(a), dev(b), ws, scal, d_a)
    torch.cuda.synchronize()
    s = scal.cpu()
    check(s[3:8], sims, "ssim levels", 2e-5)
    check(s[8:13], css, "cs levels", 2e-5)
    check(s[1], loss, "msssim loss", 2e-5)
    loss.backward()
    check(d_a.view(B, 3, 128, 128), a_ref.grad, "msssim grad", rel=True)


def test_msssim_128_wide_streamed_level_is_the_tile_kernel_bit_for_bit(tmp_path):
    """The streamed level keeps the tile kernel's item shapes and fma chains: d_recon (the F fields of all five levels through the
    backward pass) must be bit-identical to the 16 x 64 tile kernel's (CVAE_MS_STREAM=0) and the loss scalars equal to 1e-7 (the order
    of the per-plane partial sums differs).  The switch is read once per process: two child processes run
    profiles/experiments/ms_stream_check.py (B = 5 and 64) and the dumps are compared here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "profiles", "experiments", "ms_stream_check.py")
    dumps = {}
    for mode in ("1", "0"):
        out = str(tmp_path / f"ms{mode}.npz")
        env = dict(os.environ, CVAE_MS_STREAM=mode)
        r = subprocess.run([sys.executable, script, out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        dumps[mode] = np.load(out)
    for k in dumps["1"].files:
        if k.startswith("d_recon"):
            assert np.array_equal(dumps["1"][k], dumps["0"][k]), k
        else:
            assert np.abs(dumps["1"][k] - dumps["0"][k]).max() < 1e-7, k


def test_adam_matches_torch(H):
    n = 4096
    p0, g = rnd("ap", (n,)), rnd("ag", (n,), -1e-2, 1e-2)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-5)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        gs = g * step
        p_ref.grad = gs.clone()
        opt.step()
        H.adam_step(p, dev(gs * 2.0), m, v, step, 5e-5, grad_scale=0.5)
    torch.cuda.synchronize()
    check(p, p_ref, "adam params", 1e-7)


@pytest.mark.parametrize("scale", [1.0, 2.0])
def test_critic_forward_and_preprocess(H, golden_dir, scale):
    """cvae_critic_forward / cvae_preprocess_u8 vs the oracle (and the reference-generated fixture)."""
    import os
    from critic_vae_amd.critic import Critic
    fx = np.load(os.path.join(golden_dir, "critic_b8.npz"))
    cp = {k: torch.from_numpy(v) * scale for k, v in synth.make_critic_params(int(fx["wseed"])).items()}
    x, _, _ = synth.make_batch(int(fx["dseed"]), int(fx["step"]), int(fx["batch"]))
    x = torch.from_numpy(x)
    want = orc.critic_forward(cp, x)
    crit = Critic(handle=H).cuda()
    crit.load_state_dict(cp)
    got = crit.evaluate(x.cuda())
    torch.cuda.synchronize()
    check(got, want, "critic pred", 2e-6)
    if scale == 1.0:
        check(got, torch.from_numpy(fx["pred"]), "critic pred vs reference fixture", 2e-6)
    else:
        assert want.std().item() > 1e-3          # the scaled weights really exercise the network
    sd = crit.state_dict()
    assert all(torch.equal(sd[k].cpu(), cp[k]) for k in cp)
    u8 = torch.from_numpy(fx["u8"])
    pre = crit.preprocess(u8.cuda())
    torch.cuda.synchronize()
    assert torch.equal(pre.cpu(), orc.preprocess_frames(u8))
    if scale == 1.0:        # the reference's REAL checkpoint: weights + predictions of its own Critic.evaluate
        fr = np.load(os.path.join(golden_dir, "critic_real_b8.npz"))
        crit.load_state_dict({k[2:]: torch.from_numpy(fr[k]) for k in fr.files if k.startswith("w/")})
        got = crit.evaluate(x.cuda())
        torch.cuda.synchronize()
        check(got, torch.from_numpy(fr["pred"]), "critic pred, real checkpoint vs reference fixture", 2e-6)
