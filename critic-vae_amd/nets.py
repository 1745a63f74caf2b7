"""Host-side mirror of the reference's model API for the training path (vae_nets.py:7-147).

Same class / method names and call signatures as the reference so that the loop of
vae.py:44-58 runs unmodified:

    autoencoder = VariationalAutoencoder().to(device)
    opt = torch.optim.Adam(autoencoder.parameters(), lr=lr)
    out = autoencoder(images, preds)                    # (x, mu, logvar, recon)
    losses = autoencoder.vae_loss(out[0], out[1], out[2], out[3])
    losses['total_loss'].backward(); opt.step()

but every FLOP runs in libcvae_hip.so (hand-written HIP, see include/cvae.h).  All parameters
live in ONE flat fp32 nn.Parameter in the library's native layout (`theta`); `.grad` of it is the
flat gradient buffer that the data-parallel all-reduce and the fused Adam operate on.
`encoder.state_dict()` / `decoder.state_dict()` convert to the reference's key names and
layouts (vae.py:162-163), `load_state_dict` converts back.
"""
import torch
from torch import nn

from . import layout as L
from . import params as P
from . import synth
from .lib import Handle, N_SCALARS


class _ForwardFn(torch.autograd.Function):
    """cvae_forward / cvae_backward behind autograd (VariationalAutoencoder.forward, :14-19)."""

    @staticmethod
    def forward(ctx, vae, x, pred, eps, theta):
        B = x.shape[0]
        mu = torch.empty(B, P.latent_dim, device=x.device)
        logvar = torch.empty_like(mu)
        recon = torch.empty(B, P.ch, vae.width, vae.width, device=x.device)
        ws = vae._workspace(B)
        vae.handle.forward(B, x, pred, eps, theta, vae.bn_state, mu, logvar, recon, ws, train=vae.training)
        if vae.training:
            vae.num_batches_tracked += 1
        ctx.vae = vae
        ctx.ws_generation = vae._stamp_workspace()     # the saved activations live in the ONE shared workspace
        ctx.save_for_backward(x, pred, eps, theta, logvar, recon)
        ctx.mark_non_differentiable()
        return mu, logvar, recon

    @staticmethod
    def backward(ctx, d_mu, d_logvar, d_recon):
        vae = ctx.vae
        if ctx.ws_generation != vae._ws_generation:
            raise RuntimeError(
                "critic_vae_amd: backward() of a forward whose saved activations have been overwritten — another "
                "forward / encoder / decoder call of the same VariationalAutoencoder ran in between (activations are "
                "kept in one shared workspace, not per autograd graph). Run backward before the next forward, or use a "
                "second VariationalAutoencoder for evaluation.")
        x, pred, eps, theta, logvar, recon = ctx.saved_tensors
        B = x.shape[0]
        d_mu = torch.zeros_like(logvar) if d_mu is None else d_mu.contiguous()
        d_logvar = torch.zeros_like(logvar) if d_logvar is None else d_logvar.contiguous()
        d_recon = torch.zeros_like(recon) if d_recon is None else d_recon.contiguous()
        # fresh, unshared buffer so AccumulateGrad adopts it as theta.grad without a copy; the library writes
        # every element (phase bit 3 zeroes the alignment padding) — no 10 MB fill per step
        grads = torch.empty_like(theta)
        vae.handle.backward(B, x, pred, eps, theta, logvar, recon, d_recon, d_mu, d_logvar, vae._workspace(B), grads,
                            zero_padding=True)
        return None, None, None, None, grads


class _LossFn(torch.autograd.Function):
    """cvae_loss behind autograd (VariationalAutoencoder.vae_loss, :53-62)."""

    @staticmethod
    def forward(ctx, vae, x, mu, logvar, recon):
        B = x.shape[0]
        scalars = torch.empty(N_SCALARS, device=x.device)
        d_recon, d_mu, d_logvar = torch.empty_like(recon), torch.empty_like(mu), torch.empty_like(logvar)
        vae.handle.loss(B, x, mu.contiguous(), logvar.contiguous(), recon.contiguous(), vae._workspace(B), scalars,
                        d_recon, d_mu, d_logvar)
        ctx.save_for_backward(d_recon, d_mu, d_logvar)
        ctx.vae = vae
        vae.last_scalars = scalars
        return scalars[0].clone()

    @staticmethod
    def backward(ctx, g):
        d_recon, d_mu, d_logvar = ctx.saved_tensors
        o_recon, o_mu, o_logvar = torch.empty_like(d_recon), torch.empty_like(d_mu), torch.empty_like(d_logvar)
        ctx.vae.handle.scale_loss_grads(d_mu.shape[0], g.to(torch.float32).contiguous(), d_recon, d_mu, d_logvar,
                                        o_recon, o_mu, o_logvar)         # one launch, g stays on the device
        return None, None, o_mu, o_logvar, o_recon


class VariationalEncoder(nn.Module):
    """vae_nets.py:64-111.  A view onto the parent's flat parameter; owns no tensors."""

    def __init__(self, parent):
        super().__init__()
        object.__setattr__(self, "_parent", parent)

    def forward(self, x):
        return self._parent._encode(x)

    def state_dict(self, *a, **k):
        p = self._parent
        ref = L.native_to_ref(p.handle.layout, p.theta.detach())
        ref.update(L.bn_state_to_ref(p.bn_state, p.num_batches_tracked))
        keys = [f"model.{i}.{s}" for ci in L.ENC_CONV for i, s in
                ((ci, "weight"), (ci, "bias"), (ci + 1, "weight"), (ci + 1, "bias"), (ci + 1, "running_mean"),
                 (ci + 1, "running_var"), (ci + 1, "num_batches_tracked"))]
        keys += ["fc_mu.weight", "fc_mu.bias", "fc_var.weight", "fc_var.bias"]
        return {kk: ref["encoder." + kk] for kk in keys}

    def load_state_dict(self, sd, strict=True):
        self._parent._load_ref({("encoder." + k): v for k, v in sd.items()}, part="encoder")


class Decoder(nn.Module):
    """vae_nets.py:113-147."""

    def __init__(self, parent):
        super().__init__()
        object.__setattr__(self, "_parent", parent)

    def forward(self, z, pred, evalu=False, dim=1):
        if evalu:                      # vae_nets.py:140-142: batch of one, concatenate on dim 0
            z = z[0]
            dim = 0
        zcat = torch.cat((z, pred), dim=dim).reshape(-1, P.latent_dim + 1).contiguous()
        return self._parent._decode(zcat)

    def state_dict(self, *a, **k):
        p = self._parent
        ref = L.native_to_ref(p.handle.layout, p.theta.detach())
        keys = [f"model.{ci}.{s}" for ci in L.DEC_CONV for s in ("weight", "bias")]
        keys += ["decoder_input.weight", "decoder_input.bias"]
        return {kk: ref["decoder." + kk] for kk in keys}

    def load_state_dict(self, sd, strict=True):
        self._parent._load_ref({("decoder." + k): v for k, v in sd.items()}, part="decoder")


class MSSIM(nn.Module):
    """vae_nets.py:150-247; forward(img1, img2) -> 1 - MS-SSIM (gradient flows to img1)."""

    def __init__(self, parent, in_channels=3, window_size=11, size_average=True):
        super().__init__()
        object.__setattr__(self, "_parent", parent)
        assert in_channels == 3 and window_size == 11 and size_average

    def forward(self, img1, img2):
        p = self._parent
        B = img1.shape[0]
        mu0 = torch.zeros(B, P.latent_dim, device=img1.device)
        return _LossFn.apply(p, img2.contiguous(), mu0, mu0, img1)       # KLD(0,0) == 0


class VariationalAutoencoder(nn.Module):
    def __init__(self, dims=(32, 64, 128, 256), width=P.w, max_batch=256, seed=None, precision="f32", overlap_wgrad=None):
        super().__init__()
        if tuple(dims) != P.dims:
            raise ValueError("dims[3] must be 256 (view(-1,256,4,4), vae_nets.py:144) and the HIP kernels are "
                             f"instantiated for {P.dims}")
        self.width, self.max_batch = width, max_batch
        # overlap_wgrad=True: weight-gradient kernels on the library's low-priority side stream (bit-identical results).  Pays in
        # bf16 mode (+1.5 % at B = 2048: they fill the HBM-bound BatchNorm-backward stretches of the main stream), costs 3 % in
        # fp32 mode (everything is MFMA-bound there).  Off by default: with it, kernels share the chip and per-kernel times stop
        # being comparable with a serialised rocprofv3 trace; bench.py reports the rate with it as an extra key.
        overlap_wgrad = bool(overlap_wgrad)
        self.handle = Handle(width, max_batch, overlap_wgrad=overlap_wgrad, precision=precision)
        # PyTorch-default init distribution from the deterministic generator (seed None -> torch RNG seed)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        ref = {k: torch.from_numpy(v) for k, v in synth.make_params(seed, width).items()}
        self.theta = nn.Parameter(L.ref_to_native(self.handle.layout, self.handle.param_total, ref))
        bn = torch.zeros(self.handle.bn_state_floats)
        bn[L.BN_TOTAL:] = 1.0
        self.register_buffer("bn_state", bn)
        self.num_batches_tracked = 0
        self.encoder = VariationalEncoder(self)
        self.decoder = Decoder(self)
        self.mssim_loss = MSSIM(self)
        self._ws = None
        self._ws_generation = 0            # bumped by every call that writes activations into the workspace
        self.last_scalars = None

    # ---- plumbing ----
    def _stamp_workspace(self):
        self._ws_generation += 1
        return self._ws_generation

    def _workspace(self, B):
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        need = self.handle.workspace_bytes(B) // 4
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.theta.device:
            self._ws = torch.empty(self.handle.workspace_bytes(self.max_batch) // 4, device=self.theta.device)
        return self._ws

    def _load_ref(self, ref, part):
        cur = L.native_to_ref(self.handle.layout, self.theta.detach())
        for k, v in ref.items():
            if "running_" in k or "num_batches" in k:
                continue
            cur[k] = v.to(self.theta.device, torch.float32)
        with torch.no_grad():
            self.theta.copy_(L.ref_to_native(self.handle.layout, self.handle.param_total, cur, device=self.theta.device))
            for l, ci in enumerate(L.ENC_CONV):
                km, kv = f"encoder.model.{ci + 1}.running_mean", f"encoder.model.{ci + 1}.running_var"
                c = P.dims[l]
                if km in ref:
                    self.bn_state[L.BN_OFF[l]:L.BN_OFF[l] + c] = ref[km]
                if kv in ref:
                    self.bn_state[L.BN_TOTAL + L.BN_OFF[l]:L.BN_TOTAL + L.BN_OFF[l] + c] = ref[kv]
            nbt = ref.get(f"encoder.model.{L.ENC_CONV[0] + 1}.num_batches_tracked")      # one counter: the four layers step together
            if nbt is not None:
                self.num_batches_tracked = int(nbt)

    def load_reference_params(self, ref):
        """ref: {encoder./decoder.-prefixed reference key: tensor or ndarray}."""
        ref = {k: (torch.from_numpy(v) if not torch.is_tensor(v) else v) for k, v in ref.items()}
        self._load_ref(ref, part="all")

    def reference_grads(self):
        """Flat .grad converted to the reference's names/layouts (parity tests)."""
        return L.native_to_ref(self.handle.layout, self.theta.grad)

    def _prep(self, x):
        return x.to(torch.float32).contiguous()

    def _encode(self, x):
        x = self._prep(x)
        B = x.shape[0]
        mu = torch.empty(B, P.latent_dim, device=x.device)
        logvar = torch.empty_like(mu)
        z = torch.zeros_like(mu)
        pred0 = torch.zeros(B, 1, device=x.device)
        with torch.no_grad():
            self.handle.forward(B, x, pred0, z, self.theta, self.bn_state, mu, logvar, None, self._workspace(B),
                                train=self.training)
            self._stamp_workspace()
            if self.training:
                self.num_batches_tracked += 1
        return mu, logvar

    def _decode(self, zcat):
        B = zcat.shape[0]
        recon = torch.empty(B, P.ch, self.width, self.width, device=zcat.device)
        with torch.no_grad():
            self.handle.decode(B, zcat, self.theta, recon, self._workspace(B))
            self._stamp_workspace()
        return recon

    # ---- reference API ----
    def forward(self, x, pred, eps=None):
        """vae_nets.py:14-19.  `eps` (B,32) overrides the N(0,1) draw of reparametrize."""
        x = self._prep(x)
        pred = pred.to(torch.float32).reshape(x.shape[0], 1).contiguous()
        if eps is None:
            eps = torch.randn(x.shape[0], P.latent_dim, device=x.device)      # randn_like(std), :50
        mu, logvar, recon = _ForwardFn.apply(self, x, pred, eps.contiguous(), self.theta)
        return x, mu, logvar, recon

    def reparametrize(self, mu, logvar):
        """vae_nets.py:48-51 (stand-alone use; forward() fuses it into the encoder epilogue)."""
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    def vae_loss(self, x, mu, logvar, recon):
        """vae_nets.py:53-62 (no empty_cache(): the library never allocates)."""
        total = _LossFn.apply(self, self._prep(x), mu, logvar, recon)
        s = self.last_scalars
        return {"total_loss": total, "recon_loss": s[1].detach(), "KLD": s[2].detach()}

    def recon_samples(self, x, reward):
        """vae_nets.py:21-29."""
        mu, logvar = self.encoder(x)
        return [self.decoder(self.reparametrize(mu, logvar), reward) for _ in range(6)]

    def inject(self, x, reward=None):
        """vae_nets.py:31-40."""
        if reward is None:
            reward = torch.tensor([0, 0.2, 0.4, 0.6, 0.8, 1.0])
        reward = reward.to(x.device)
        mu, _ = self.encoder(x)
        return [self.decoder(mu, reward[i].view(1), evalu=True) for i in range(P.inject_n)]

    def evaluate(self, x, pred):
        """vae_nets.py:42-46."""
        mu, _ = self.encoder(x)
        return self.decoder(mu, pred.view(1), evalu=True)

    def diff_images(self, x, pred, one=False):
        """get_diff_image (vae_utility.py:256-277) for a whole batch at once: the encoder runs once,
        the decoder twice (critic value `pred` — or 1 if `one` — against 0); returns
        (recon_one, recon_zero, diff (B,w,w), per-image max of diff)."""
        x = self._prep(x)
        B = x.shape[0]
        mu, _ = self.encoder(x)
        hi = torch.ones(B, 1, device=x.device) if one else pred.to(torch.float32).reshape(B, 1)
        recon_one = self.decoder(mu, hi)
        recon_zero = self.decoder(mu, torch.zeros(B, 1, device=x.device))
        diff = torch.empty(B, self.width, self.width, device=x.device)
        self.handle.diff_grey(B, recon_one, recon_zero, diff)
        return recon_one, recon_zero, diff, diff.flatten(1).max(dim=1).values
