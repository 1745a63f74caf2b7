"""GPU parity of the whole training step (forward + vae_loss + backward [+ Adam]) through the
reference-shaped Python API, against (a) the committed reference-generated fixtures and (b) the
oracle run here on the same seeded inputs.  Tolerance 1e-4 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from critic_vae_amd import synth
from critic_vae_amd.nets import VariationalAutoencoder
from critic_vae_amd.train import FusedTrainer, train
from oracle import cvae_oracle as orc
from decisions import check_step_against_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _inputs(dseed, step, B, first=0, width=64):
    x, pred, eps = synth.make_batch(dseed, step, B, width, first_index=first)
    return torch.from_numpy(x), torch.from_numpy(pred), torch.from_numpy(eps)


def _model(B, wseed=0, width=64, precision="f32"):
    assert torch.cuda.is_available()
    vae = VariationalAutoencoder(max_batch=B, seed=wseed, width=width, precision=precision).cuda()
    vae.load_reference_params(synth.make_params(wseed, width))
    return vae


def _step(vae, x, pred, eps):
    vae.theta.grad = None
    out = vae(x.cuda(), pred.cuda(), eps=eps.cuda())
    losses = vae.vae_loss(*out)
    losses["total_loss"].backward()
    torch.cuda.synchronize()
    return out, losses


@pytest.mark.parametrize("tag", ["b2", "b32", "w128_b2"])
def test_step_matches_reference_fixture_and_oracle(golden_dir, tag):
    """w128_b2 = BASELINE.json config 5 frame size (128x128; reference patched as in make_golden.py)."""
    fx = np.load(os.path.join(golden_dir, f"step_{tag}.npz"))
    B, W = int(fx["batch"]), int(fx["width"])
    x, pred, eps = _inputs(int(fx["dseed"]), int(fx["step"]), B, width=W)
    vae = _model(B, int(fx["wseed"]), W)
    (_, mu, logvar, recon), losses = _step(vae, x, pred, eps)
    # --- against the reference-generated fixture ---
    mu, logvar, recon = mu.detach(), logvar.detach(), recon.detach()
    assert np.abs(mu.cpu().numpy() - fx["mu"]).max() < TOL
    assert np.abs(logvar.cpu().numpy() - fx["logvar"]).max() < TOL
    assert np.abs(recon.cpu().numpy().reshape(-1)[::16] - fx["recon_sample"]).max() < TOL
    s = vae.last_scalars.cpu().numpy()
    assert np.abs(s[:3] - fx["losses"]).max() < TOL
    assert np.abs(s[3:8] - fx["ssim_levels"]).max() < TOL and np.abs(s[8:13] - fx["cs_levels"]).max() < TOL
    assert abs(losses["recon_loss"].item() - fx["losses"][1]) < TOL and abs(losses["KLD"].item() - fx["losses"][2]) < TOL
    ref_g = vae.reference_grads()
    for name, g in ref_g.items():
        gs = g.cpu().numpy().reshape(-1)
        assert np.abs(gs[fx["grad_idx/" + name]] - fx["grad_val/" + name]).max() <= TOL, name
    sd = vae.encoder.state_dict()
    for bi in (1, 5, 9, 13):
        assert np.abs(sd[f"model.{bi}.running_mean"].cpu().numpy() - fx[f"bn_running_mean/{bi}"]).max() < 1e-5
        assert np.abs(sd[f"model.{bi}.running_var"].cpu().numpy() - fx[f"bn_running_var/{bi}"]).max() < 1e-5
    # --- against the oracle, every element of every gradient (abs 1e-4; 1e-4 of each tensor's max) ---
    _, o = check_step_against_oracle(vae, x, pred, eps, B, wseed=int(fx["wseed"]))
    assert (recon.cpu() - o["recon"]).abs().max().item() < TOL


@pytest.mark.parametrize("precision", ["f32", "bf16x9", "bf16x6", "bf16"])
def test_step_on_the_references_real_frames(golden_dir, precision):
    """Parity on frames that look like frames (tests/golden/step_real_b68.npz, written by the reference itself from its
    source-images/*.jpg): uint8 HWC frames on the HOST -> FrameFeeder (pinned staging, H2D, cvae_preprocess_u8) -> HIP critic
    with the reference's real checkpoint -> forward + vae_loss + backward, against the fixture at 1e-4 and, element by
    element, against the oracle with the HIP decisions imposed.  Real frames have flat regions (sky, inventory bar): max-pool
    windows near ties and ReLU inputs near zero are common here, which the U[0,1) fixtures never exercise — the flip count is
    printed.  The fp32-emulation modes (bf16x9 / bf16x6) are held to the fp32 branch in full: fixture at 1e-4, every gradient element with the
    decisions imposed, the finite-loss / NaN-gradient half.  bf16 mode: the documented bounds of test_gpu_bf16 (outputs 3e-2, loss 2e-3),
    finite gradients.
    Second half: the plain seed-0 weights, on which the REFERENCE's loss is finite but all its gradients are NaN (a negative
    ssim level under the unused `mssim ** weights`, make_golden.real_frames_case) — the HIP path must return the same."""
    from critic_vae_amd.critic import Critic
    from critic_vae_amd.feeder import FrameFeeder
    from test_oracle import real_frames_params
    fx = np.load(os.path.join(golden_dir, "step_real_b68.npz"))
    cw = np.load(os.path.join(golden_dir, "critic_real_b8.npz"))
    B = int(fx["batch"])
    assert B == 68 and fx["u8"].shape == (68, 64, 64, 3)
    params = real_frames_params(fx)
    vae = _model(B, int(fx["wseed"]), precision=precision)
    vae.load_reference_params(params)
    critic = Critic(handle=vae.handle).cuda()
    critic.load_state_dict({k[2:]: torch.from_numpy(cw[k]) for k in cw.files if k.startswith("w/")})
    feeder = FrameFeeder(fx["u8"], B, "cuda:0", vae.handle, critic=critic)
    (x, pred), = list(feeder.batches([np.arange(B)]))
    x, pred = x.clone(), pred.clone()
    assert torch.equal(x.cpu(), orc.preprocess_frames(torch.from_numpy(fx["u8"])))          # bit-exact pre-processing
    assert np.abs(pred.cpu().numpy() - fx["pred"]).max() < TOL
    eps = torch.from_numpy(synth.make_batch(int(fx["dseed"]), int(fx["step"]), B)[2])
    (_, mu, logvar, recon), losses = _step(vae, x, pred, eps)
    mu, logvar, recon, s = mu.detach().cpu().numpy(), logvar.detach().cpu().numpy(), recon.detach().cpu().numpy(), vae.last_scalars.cpu().numpy()
    exact = precision != "bf16"                      # f32 and its two emulations: the 1e-4 bar
    otol, ltol = (TOL, TOL) if exact else (3e-2, 2e-3)
    assert np.abs(mu - fx["mu"]).max() < otol and np.abs(logvar - fx["logvar"]).max() < otol
    assert np.abs(recon.reshape(-1)[::16] - fx["recon_sample"]).max() < otol
    assert np.abs(s[:3] - fx["losses"]).max() < ltol and np.isfinite(s[:13]).all()
    assert np.abs(s[8:12] - fx["cs_levels"][:4]).max() < (TOL if exact else 5e-3)
    g = vae.reference_grads()
    assert all(bool(torch.isfinite(t).all()) for t in g.values())
    if exact:
        for name, t in g.items():
            assert np.abs(t.cpu().numpy().reshape(-1)[fx["grad_idx/" + name]] - fx["grad_val/" + name]).max() <= TOL, name
        # every gradient element against the oracle run on the SAME preds the step used, HIP decisions imposed
        p = orc.to_torch(params, requires_grad=True)
        taps = {}
        o = orc.train_step(p, x.cpu(), pred.cpu(), eps, bn_state=orc.new_bn_state(p), taps=taps)
        from decisions import hip_decisions, oracle_decisions, flips, is_pre_bn_bias, tie_tol, value_deviation
        d_hip = hip_decisions(vae, B)
        fl = flips(d_hip, oracle_decisions(taps), taps)
        n_flips = sum(f[1] for f in fl)
        devn = value_deviation(vae, B, taps)
        assert all(gap <= tie_tol(vae) for _, _, gap in fl), (fl, devn)
        p2 = orc.to_torch(params, requires_grad=True)
        orc.train_step(p2, x.cpu(), pred.cpu(), eps, bn_state=orc.new_bn_state(p2), decisions=d_hip)
        worst = 0.0
        for k, w in p2.items():
            if is_pre_bn_bias(k):
                continue
            e, scale = float((g[k].cpu().double() - w.grad.double()).abs().max()), max(float(w.grad.abs().max()), 1e-30)
            worst = max(worst, e / scale)
            assert e <= 1e-4 * scale, f"{k}: {e:.3e} vs max|g| {scale:.3e} ({n_flips} flips)"
        print(f"real frames, B = {B}, {precision}: {n_flips} decision flips {fl}; rel(decisions imposed) {worst:.2e}")
        assert (torch.from_numpy(recon) - o["recon"]).abs().max().item() < TOL
    else:
        for name in ("decoder.model.12.weight", "decoder.decoder_input.weight", "encoder.fc_mu.weight"):
            a, b = g[name].cpu().numpy().reshape(-1)[fx["grad_idx/" + name]], fx["grad_val/" + name]
            assert np.abs(a - b).max() <= 0.05 * float(fx["grad_max/" + name]) + 1e-6, name
    # ---- plain seed-0 weights: finite loss, every gradient NaN — in the reference, so here ----
    vae.load_reference_params(synth.make_params(int(fx["wseed"])))
    (_, mu0, _, _), _ = _step(vae, x, pred, eps)
    s0 = vae.last_scalars.cpu().numpy()
    assert np.isfinite(s0[:3]).all() and np.abs(s0[:3] - fx["seed0/losses"]).max() < ltol
    assert s0[3] < 0 and np.abs(s0[3:8] - fx["seed0/ssim_levels"]).max() < (TOL if exact else 5e-3)
    assert np.abs(mu0.detach().cpu().numpy() - fx["seed0/mu"]).max() < otol
    g0 = vae.reference_grads()
    for k, fin in zip(fx["seed0/grad_names"], fx["seed0/grad_finite"]):
        assert bool(torch.isfinite(g0[str(k)]).all()) == bool(fin), f"{k}: the reference's gradient is {'finite' if fin else 'NaN'}"


def test_intermediates_against_oracle_taps():
    """Every saved activation and activation-gradient of the step vs the oracle's taps (B=4)."""
    B = 4
    x, pred, eps = _inputs(1234, 3, B)
    vae = _model(B)
    _step(vae, x, pred, eps)
    p = orc.to_torch(synth.make_params(0), requires_grad=True)
    taps = {}
    orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p), taps=taps)
    ws, h = vae._workspace(B), vae.handle
    enc = [(32, 64), (64, 32), (128, 16), (256, 8)]
    dec = [(128, 4), (64, 8), (32, 16), (32, 32)]

    def view(name, c, s):
        return h.ws_view(ws, B, name, B * s * s * c).view(B, s, s, c).permute(0, 3, 1, 2).cpu()

    for l, (c, s) in enumerate(enc):
        assert (view(f"y{l}", c, s) - taps[f"enc_y{l}"]).abs().max() < TOL, f"y{l}"
        assert (view(f"a{l}", c, s // 2) - taps[f"enc_a{l}"]).abs().max() < TOL, f"a{l}"
    assert (view("h", 256, 4) - taps["dec_h"]).abs().max() < TOL
    for i, (c, s) in enumerate(dec):
        assert (view(f"o{i}", c, s) - taps[f"dec_o{i}"]).abs().max() < TOL, f"o{i}"
    from ws_tools import recompute_d_y0
    d_y0 = recompute_d_y0(h, ws, B)  # the step applies block 0's BatchNorm backward inside E1's weight-gradient kernel: no d_y0 slot
    assert h.lib.cvae_ws_offset(h.h, B, b"d_y0") < 0
    for l, (c, s) in enumerate(enc):
        g = taps[f"enc_y{l}"].grad
        got = d_y0.view(B, s, s, c).permute(0, 3, 1, 2).cpu() if l == 0 else view(f"d_y{l}", c, s)
        assert (got - g).abs().max() <= TOL * max(1.0, 0) + 1e-4 * g.abs().max(), f"d_y{l}"


@pytest.mark.parametrize("prec,width,rel", [("f32", 64, 1e-6), ("f32", 128, 1e-6), ("bf16", 64, 2e-2)])
def test_fused_e1_backward_equals_the_separate_apply_pass(monkeypatch, prec, width, rel):
    """Block 0's BatchNorm / pool / ReLU backward runs inside E1's weight-gradient kernel (E1Fuse, the default) or as
    bn.hip's apply pass that writes d_y0 (CVAE_FUSE_E1=0).  fp32: the same arithmetic per element, so dW1 / db1 agree to
    fp32 summation noise; bf16: the fused form folds the per-channel constants before the bf16 rounding of dy."""
    from critic_vae_amd import layout as L
    B = 8
    x, pred, eps = (t.cuda() for t in _inputs(1234, 0, B, width=width))
    grads = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("CVAE_FUSE_E1", fuse)
        vae = VariationalAutoencoder(max_batch=B, seed=0, width=width, precision=prec).cuda()
        vae.load_reference_params(synth.make_params(0, width))
        tr = FusedTrainer(vae)
        tr.step(x, pred, eps)
        torch.cuda.synchronize()
        grads[fuse] = {k: v.clone() for k, v in L.native_to_ref(vae.handle.layout, tr.grads.cpu()).items()}
    for k in grads["1"]:
        a, b = grads["1"][k], grads["0"][k]
        if k.startswith("encoder.model.0."):
            bound = rel * b.abs().max().item() + (1e-6 if k.endswith("bias") else 0.0)      # db1 cancels to ~0
            assert (a - b).abs().max().item() <= bound, (k, (a - b).abs().max().item(), b.abs().max().item())
        else:
            assert torch.equal(a, b), k           # nothing else may change


@pytest.mark.parametrize("precision", ["f32", "bf16x9", "bf16x6"])
def test_step_b256_fp32_against_oracle(precision):
    """BASELINE.json configs[1] — the bench workload — at full size: fp32, B=256, vs the CPU oracle at 1e-4
    on mu / logvar / recon / the loss scalars and on EVERY gradient element; and at 1e-4 of each gradient
    tensor's max once the handful of max-pool / ReLU decisions that sit inside fp32 round-off of a tie are
    imposed on the oracle (tests/decisions.py: they are counted, and each is shown to be a tie).
    The fp32-emulation modes (exact 3-way bf16 operand splits on the bf16 MFMA: "bf16x9" all nine partial products,
    "bf16x6" the six leading ones) are held to exactly the same bar — same test, same tolerances."""
    B = 256
    x, pred, eps = _inputs(1234, 0, B)
    vae = _model(B, precision=precision)
    (_, mu, logvar, recon), losses = _step(vae, x, pred, eps)
    rep, o = check_step_against_oracle(vae, x, pred, eps, B, verbose=True)
    assert rep is not None, "seed must give a finite loss"
    assert (mu.detach().cpu() - o["mu"]).abs().max() < TOL and (logvar.detach().cpu() - o["logvar"]).abs().max() < TOL
    assert (recon.detach().cpu() - o["recon"]).abs().max() < TOL
    s = vae.last_scalars.cpu()
    assert abs(float(s[0]) - float(o["total_loss"])) < TOL and abs(float(s[1]) - float(o["recon_loss"])) < TOL
    assert abs(float(s[2]) - float(o["KLD"])) < TOL
    assert (s[3:8] - o["ssim_levels"]).abs().max() < TOL and (s[8:13] - o["cs_levels"]).abs().max() < TOL
    assert rep["rel_forced"] <= 1e-4


def test_bitwise_reproducible_and_full_size():
    """BASELINE config 2 size (B=256): finite, and two runs are bit-identical (fixed-order reductions)."""
    B = 256
    x, pred, eps = _inputs(1234, 0, B)
    vae = _model(B)
    (_, mu, _, recon), losses = _step(vae, x, pred, eps)
    g1, l1 = vae.theta.grad.clone(), vae.last_scalars.clone()
    assert torch.isfinite(l1[:13]).all() and torch.isfinite(g1).all()
    assert recon.abs().max().item() <= 1.0 and mu.shape == (B, 32)
    _step(vae, x, pred, eps)
    assert torch.equal(g1, vae.theta.grad) and torch.equal(l1, vae.last_scalars)
    # linearity of backward in the loss gradient: 2*loss -> 2*grads
    vae.theta.grad = None
    out = vae(x.cuda(), pred.cuda(), eps=eps.cuda())
    (2.0 * vae.vae_loss(*out)["total_loss"]).backward()
    torch.cuda.synchronize()
    assert (vae.theta.grad - 2 * g1).abs().max().item() <= 2e-6 * g1.abs().max().item() + 1e-9


def test_trajectory_config1_fused_trainer(golden_dir):
    """BASELINE config 1 on the GPU path: 32 Adam steps, B=32, against the reference trajectory."""
    fx = np.load(os.path.join(golden_dir, "trajectory_b32.npz"))
    B = int(fx["batch"])
    vae = _model(B, int(fx["wseed"]))
    tr = FusedTrainer(vae)
    got = []
    for s in range(int(fx["n_frames"]) // B):
        x, pred, eps = _inputs(int(fx["dseed"]), s, B)
        got.append(tr.step(x.cuda(), pred.cuda(), eps.cuda())[:3].cpu().numpy().copy())
    got = np.array(got)
    assert np.isfinite(got).all()
    assert np.abs(got[:2] - fx["traj"][:2]).max() < TOL                 # before Adam noise can amplify
    assert np.abs(got - fx["traj"]).max() < 5e-3, np.abs(got - fx["traj"]).max()
    assert got[-1, 0] < 0.5 * got[0, 0]                                 # it trains


@pytest.mark.parametrize("precision", ["f32", "bf16x6"])
def test_batchnorm_statistics_of_channels_far_from_zero(precision):
    """the persistent conv kernels take a tile's BatchNorm M2 as Q - S^2 / n from single-pass fp32 sums of the BIASED accumulators;
    its relative error grows like eps * sqrt(n) * (mean^2 / var + 1), and the fixtures' channels have |mean| / std ~ 1.  Here the conv biases of
    blocks 1..3 push every channel's mean to ~30 standard deviations: the running variance (what the statistics end up as) and the step's outputs
    must still meet the oracle — variance within 1e-3 of its value, outputs and loss at 1e-4."""
    B = 8
    x, pred, eps = _inputs(1234, 5, B)
    params = synth.make_params(0)
    for k in ("encoder.model.4.bias", "encoder.model.8.bias", "encoder.model.12.bias"):
        params[k] = params[k] + np.float32(12.0)          # y of these layers has std ~0.4 on the seed-0 weights: mean / std ~ 30
    vae = _model(B, 0, precision=precision)
    vae.load_reference_params(params)
    (_, mu, logvar, recon), losses = _step(vae, x, pred, eps)
    p = orc.to_torch(params, requires_grad=True)
    bn = orc.new_bn_state(p)
    taps = {}
    o = orc.train_step(p, x, pred, eps, bn_state=bn, taps=taps)
    sd = vae.encoder.state_dict()
    for l, bi in enumerate((1, 5, 9, 13)):
        y = taps[f"enc_y{l}"].detach()
        ratio = float((y.mean(dim=(0, 2, 3)).abs() / y.std(dim=(0, 2, 3))).median())
        rv, want = sd[f"model.{bi}.running_var"].cpu(), bn[f"encoder.model.{bi}.running_var"]
        rel = float(((rv - want).abs() / want.abs().clamp_min(1e-12)).max())
        print(f"block {l}: median |mean| / std {ratio:.1f}; running_var rel err {rel:.2e}")
        assert rel < 1e-3, (l, rel)
        if l > 0:
            assert ratio > 20
    assert (mu.detach().cpu() - o["mu"].detach()).abs().max() < TOL and (recon.detach().cpu() - o["recon"].detach()).abs().max() < TOL
    assert abs(float(losses["total_loss"].item()) - float(o["total_loss"].detach())) < TOL


def test_training_on_the_references_real_frames_fp32_and_bf16(golden_dir):
    """Training on frames that look like frames: 200 Adam steps (lr 5e-5, the reference's) on the 68 real frames as ONE batch, from the weights of
    step_real_b68.npz, eps of step s from the generator — against tests/golden/train_real_b68.npz, written by the REFERENCE's own modules +
    torch.optim.Adam (make_golden.real_frames_training_case; its loss stays finite for all 200 steps: 0.32773 -> 0.13144, so the whole run is
    the finite prefix).  fp32 mode: the first two steps at 1e-4 (before Adam can amplify round-off), the whole curve within 5e-3 as the
    config-1 trajectory test.  bf16 mode (bf16 MFMA + bf16 activation storage, not held to 1e-4): the same run from the same eps stream must stay with
    the fp32-mode curve — max gap and final ratio bounded as test_bf16_config2_trains_like_fp32_at_full_batch does on noise frames, with the bounds
    scaled for B = 68 instead of 2048 (the bf16 gradient noise is ~1/sqrt(B): measured values printed)."""
    from test_oracle import real_frames_params
    fx = np.load(os.path.join(golden_dir, "step_real_b68.npz"))
    tf = np.load(os.path.join(golden_dir, "train_real_b68.npz"))
    assert int(tf["first_nonfinite_step"]) == -1 and np.isfinite(tf["traj"]).all()
    B, steps = 68, int(tf["steps"])
    x = orc.preprocess_frames(torch.from_numpy(fx["u8"])).cuda()
    pred = torch.from_numpy(fx["pred"]).cuda()
    curves = {}
    for prec in ("f32", "bf16"):
        vae = _model(B, int(fx["wseed"]), precision=prec)
        vae.load_reference_params(real_frames_params(fx))
        tr = FusedTrainer(vae)
        got = torch.empty(steps, 3, device="cuda")
        for s_ in range(steps):
            eps = torch.from_numpy(synth.make_batch(int(tf["dseed"]), s_, B)[2]).cuda()
            got[s_] = tr.step(x, pred, eps)[:3]
        curves[prec] = got.cpu().numpy()
        del tr, vae
    a, b, r = curves["bf16"], curves["f32"], tf["traj"]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    d01, dall = np.abs(b[:2] - r[:2]).max(), np.abs(b - r).max()
    gap, ratio = np.abs(a[:, 0] - b[:, 0]).max(), a[-8:, 0].mean() / b[-8:, 0].mean()
    print(f"real frames, {steps} steps at B = {B}: reference {r[0, 0]:.5f} -> {r[-1, 0]:.5f}; f32 mode {b[-1, 0]:.5f} (first two steps {d01:.1e}, whole curve {dall:.1e}); "
          f"bf16 mode {a[-1, 0]:.5f}: max gap to f32 mode {gap:.2e}, final ratio {ratio:.5f}")
    assert d01 < TOL and dall < 5e-3
    assert b[-1, 0] < 0.5 * b[0, 0] and a[-1, 0] < 0.5 * a[0, 0]          # both train
    assert gap < 2e-3 and abs(ratio - 1.0) < 4e-3          # measured: max gap 5.6e-4, final ratio 1.0013 (fp32 mode vs the reference: 2.8e-4 over the run)


def test_reference_loop_is_a_drop_in():
    """vae.py:33-66 verbatim (torch.optim.Adam over .parameters(), tail batch kept) == fused trainer."""
    B, n = 32, 80                       # 80 frames -> batches of 32, 32, 16 (short tail)
    frames, _, _ = synth.make_batch(1234, 0, n)
    preds_all = torch.from_numpy(synth.uniform(3, "critic", (n, 1)))
    vae_a, vae_b = _model(B), _model(B)
    np.random.seed(0)
    torch.manual_seed(0)
    order = []

    def critic(images):                 # deterministic stand-in for critic.evaluate (vae.py:50)
        idx = [int(np.argmin(np.abs(frames.reshape(n, -1)[:, :8] - im.reshape(-1)[:8].cpu().numpy()).sum(1))) for im in images]
        order.append(idx)
        return preds_all[idx].cuda()

    _, hist = train(vae_a, [f[None] for f in frames], critic, torch.device("cuda"), epochs=1, batch_size=B, log_n=B)
    assert len(order) == 3 and len(order[-1]) == 16
    # replay the same batches / noise through the fused trainer
    torch.manual_seed(0)
    tr = FusedTrainer(vae_b)
    for idx in order:
        xb = torch.from_numpy(frames[idx]).cuda()
        eps = torch.randn(len(idx), 32, device="cuda")
        tr.step(xb, preds_all[idx].cuda(), eps)
    torch.cuda.synchronize()
    # Same kernels, same gradients at step 0; torch's foreach-Adam and the fused Adam differ in the
    # last ulp of m/v, and Adam turns round-off-level gradients (pre-BatchNorm biases, dead units)
    # into O(lr) moves -> bound by lr*steps, and require that almost all agree.
    diff = (vae_a.theta - vae_b.theta).detach().abs()
    assert diff.max().item() < 3 * 5e-5 and diff.mean().item() < 2e-7
    assert all(np.isfinite(list(r.values())).all() for _, r in hist)


def test_state_dict_round_trip_reference_layout():
    vae = _model(4)
    ref = synth.make_params(0)
    enc, dec = vae.encoder.state_dict(), vae.decoder.state_dict()
    for k, v in ref.items():
        part, key = k.split(".", 1)
        got = (enc if part == "encoder" else dec)[key].cpu().numpy()
        assert got.shape == v.shape and np.array_equal(got, v), k
    assert enc["model.1.num_batches_tracked"].item() == 0
    vae2 = VariationalAutoencoder(max_batch=4, seed=5).cuda()
    vae2.encoder.load_state_dict(enc)
    vae2.decoder.load_state_dict(dec)
    assert torch.equal(vae2.theta, vae.theta)


def test_train_main_save_writes_the_two_reference_files(tmp_path):
    """vae.py:162-163: `-train` ends with torch.save(vae.encoder.state_dict(), ENCODER_PATH) / the decoder's.  `train.main
    --save DIR` writes DIR/vae_encoder.pt and DIR/vae_decoder.pt in the reference's key names, shapes and dtypes; loaded
    back (load_networks = load_vae_network, vae_utility.py:345-361) they reproduce the trained flat parameter and the
    BatchNorm running statistics bit for bit."""
    from critic_vae_amd import train as T
    T.main(["-train", "--synthetic", "64", "--batch", "16", "--epochs", "1", "--save", str(tmp_path)])
    enc = torch.load(tmp_path / "vae_encoder.pt")
    dec = torch.load(tmp_path / "vae_decoder.pt")
    ref = synth.make_params(0)
    assert sorted(enc) == sorted(k.split(".", 1)[1] for k in _ref_keys("encoder"))
    assert sorted(dec) == sorted(k.split(".", 1)[1] for k in _ref_keys("decoder"))
    for k, v in ref.items():                       # reference shapes (OIHW conv weights, fc (32, 4096), decoder_input (4096, 33))
        part, key = k.split(".", 1)
        got = (enc if part == "encoder" else dec)[key]
        assert tuple(got.shape) == v.shape and got.dtype == torch.float32 and not got.is_cuda, k
    assert enc["model.1.num_batches_tracked"].item() == 4 and enc["model.1.num_batches_tracked"].dtype == torch.int64
    assert not np.array_equal(enc["model.0.weight"].numpy(), ref["encoder.model.0.weight"])       # it trained
    # the same run again, in-process, and the files loaded into a fresh model: identical parameters and statistics
    torch.manual_seed(0); np.random.seed(0)
    vae = VariationalAutoencoder(max_batch=16, seed=0).cuda()
    T.train(vae, T.synthetic_dataset(64), lambda im: torch.rand(im.shape[0], 1, device=im.device), "cuda:0",
            epochs=1, batch_size=16, log_n=16 * 8, log=lambda m: None)
    vae2 = T.load_networks(VariationalAutoencoder(max_batch=16, seed=5).cuda(), str(tmp_path))
    assert torch.equal(vae2.theta, vae.theta) and torch.equal(vae2.bn_state, vae.bn_state)
    assert int(vae2.num_batches_tracked) == 4


def _ref_keys(part):
    keys = [k for k in synth.make_params(0) if k.startswith(part + ".")]
    if part == "encoder":
        keys += [f"encoder.model.{i}.{s}" for i in (1, 5, 9, 13) for s in ("running_mean", "running_var", "num_batches_tracked")]
    return keys


def test_inference_path_eval_mode(golden_dir):
    """evaluate / inject / diff-mask (vae_nets.py:31-46, vae_utility.py:256-277) in
    eval mode (BatchNorm running statistics, mu instead of a sample), batched, vs the oracle AND vs the
    fixture written by the reference's own evaluate / inject (tests/golden/inference_b5.npz)."""
    fx = np.load(os.path.join(golden_dir, "inference_b5.npz"))
    B = int(fx["batch"])
    assert B == 5 and list(fx["train_steps"]) == [20, 21, 22] and int(fx["step"]) == 9
    x, pred, eps = _inputs(1234, 9, B)
    vae = _model(B)
    # a few training steps give non-trivial running statistics on both sides
    p = orc.to_torch(synth.make_params(0), requires_grad=True)
    bn = orc.new_bn_state(p)
    for s in range(3):
        xs, ps, es = _inputs(1234, 20 + s, B)
        _step(vae, xs, ps, es)
        orc.zero_grad(p)
        orc.train_step(p, xs, ps, es, bn_state=bn)
    vae.eval()
    with torch.no_grad():
        mu, logvar = vae.encoder(x.cuda())
        mu_o, lv_o = orc.encoder(p, x, bn, train=False)
        assert (mu.cpu() - mu_o).abs().max() < TOL and (logvar.cpu() - lv_o).abs().max() < TOL
        r1, r0, diff, mx = vae.diff_images(x.cuda(), pred.cuda())
        r1_o = orc.decoder(p, mu_o, pred)
        r0_o = orc.decoder(p, mu_o, torch.zeros(B, 1))
        assert (r1.cpu() - r1_o).abs().max() < TOL and (r0.cpu() - r0_o).abs().max() < TOL
        d_o = ((r0_o - r1_o).abs() * torch.tensor([0.2989, 0.5870, 0.1140]).view(1, 3, 1, 1)).sum(1)
        assert (diff.cpu() - d_o).abs().max() < TOL and (mx.cpu() - d_o.flatten(1).max(1).values).abs().max() < TOL
        # single-image API exactly as the reference calls it (batch of one, pred.view(1))
        one = vae.evaluate(x[:1].cuda(), pred[0].cuda())
        assert one.shape == (1, 3, 64, 64) and (one.cpu() - r1_o[:1]).abs().max() < TOL
        inj = vae.inject(x[:1].cuda())
        assert len(inj) == 6 and (inj[0].cpu() - r0_o[:1]).abs().max() < TOL
        # --- the reference's own numbers ---
        assert np.abs(mu.cpu().numpy() - fx["mu"]).max() < TOL and np.abs(logvar.cpu().numpy() - fx["logvar"]).max() < TOL
        assert np.abs(r1.cpu().numpy().reshape(B, -1)[:, ::4] - fx["evaluate_pred_sample"]).max() < TOL
        assert np.abs(r0.cpu().numpy().reshape(B, -1)[:, ::4] - fx["evaluate_zero_sample"]).max() < TOL
        got_inj = np.stack([r.cpu().numpy().reshape(-1)[::4] for r in inj])
        assert np.abs(got_inj - fx["inject_first_frame_sample"]).max() < TOL
        assert np.abs(mx.cpu().numpy() - fx["diff_max"]).max() < TOL
        for i in range(B):                                   # per frame, exactly as get_diff_image calls it
            one_i = vae.evaluate(x[i:i + 1].cuda(), pred[i].cuda())
            assert np.abs(one_i.cpu().numpy().reshape(-1)[::4] - fx["evaluate_pred_sample"][i]).max() < TOL
    sd = vae.encoder.state_dict()
    for bi in (1, 5, 9, 13):
        assert np.abs(sd[f"model.{bi}.running_mean"].cpu().numpy() - fx[f"bn_running_mean/{bi}"]).max() < 1e-5
        assert np.abs(sd[f"model.{bi}.running_var"].cpu().numpy() - fx[f"bn_running_var/{bi}"]).max() < 1e-5
    # eval mode must not touch the running statistics
    before = vae.bn_state.clone()
    with torch.no_grad():
        vae.encoder(x.cuda())
    assert torch.equal(before, vae.bn_state)


@pytest.mark.parametrize("B", [1, 7])
def test_ragged_batches_against_oracle(B):
    """Tail batches of any size are kept by the reference loop (vae.py:44-46): B=1 (BatchNorm over one
    image), B=7 (partial 2- and 8-image tiles), with a workspace sized for a larger max_batch."""
    x, pred, eps = _inputs(1234, 40 + B, B)
    assert torch.cuda.is_available()
    vae = VariationalAutoencoder(max_batch=16, seed=0).cuda()
    vae.load_reference_params(synth.make_params(0))
    (_, mu, logvar, recon), losses = _step(vae, x, pred, eps)
    p = orc.to_torch(synth.make_params(0), requires_grad=True)
    o = orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p))
    if not torch.isfinite(o["total_loss"]):
        assert not torch.isfinite(losses["total_loss"]).item()      # NaN propagates identically
        return
    assert (mu.detach().cpu() - o["mu"]).abs().max() < TOL and (recon.detach().cpu() - o["recon"]).abs().max() < TOL
    assert abs(losses["total_loss"].item() - o["total_loss"].item()) < TOL
    check_step_against_oracle(vae, x, pred, eps, B)
