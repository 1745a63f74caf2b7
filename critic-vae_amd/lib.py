"""ctypes binding of libcvae_hip.so (C-ABI in include/cvae.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.  PyTorch is
used only as the owner of device memory and streams; raw pointers cross the boundary.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVAE_LIB") or os.path.join(_HERE, "libcvae_hip.so")     # CVAE_LIB: A/B another build of the same sources
N_SCALARS = 16


class CvaeError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("max_batch", C.c_int32),
                ("overlap_wgrad", C.c_int32), ("precision", C.c_int32)]


def build(verbose=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc")],
                       capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise CvaeError("building libcvae_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


_lib = None
_p, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float

_SIGS = {
    "cvae_version": (C.c_char_p, []),
    "cvae_last_error": (C.c_char_p, []),
    "cvae_create": (C.c_int, [C.POINTER(_Config), C.POINTER(_p)]),
    "cvae_destroy": (None, [_p]),
    "cvae_param_total": (_i64, [_p]),
    "cvae_param_count": (_i32, [_p]),
    "cvae_param_name": (C.c_char_p, [_p, _i32]),
    "cvae_param_offset": (_i64, [_p, _i32]),
    "cvae_param_numel": (_i64, [_p, _i32]),
    "cvae_workspace_bytes": (_i64, [_p, _i32]),
    "cvae_bn_state_floats": (_i64, [_p]),
    "cvae_ws_offset": (_i64, [_p, _i32, C.c_char_p]),
    "cvae_conv_route": (_i32, [_i32, _i32, _i32, _i32, _i64]),
    "cvae_forward": (C.c_int, [_p, _i32] + [_p] * 9 + [_i32, _p]),
    "cvae_decode": (C.c_int, [_p, _i32] + [_p] * 5),
    "cvae_loss": (C.c_int, [_p, _i32] + [_p] * 10),
    "cvae_backward": (C.c_int, [_p, _i32] + [_p] * 12),
    "cvae_backward_phases": (C.c_int, [_p, _i32] + [_p] * 11 + [_i32, _p]),
    "cvae_grad_bucket": (C.c_int, [_p, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "cvae_scale_loss_grads": (C.c_int, [_p, _i32] + [_p] * 8),
    "cvae_adam_step": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f, _f, _f, _f, _f, _p]),
    "cvae_grads_to_bf16": (C.c_int, [_p, _p, _p, _i64, _p]),
    "cvae_grads_from_bf16": (C.c_int, [_p, _p, _p, _i64, _p]),
    "cvae_critic_param_count": (_i32, []),
    "cvae_critic_forward": (C.c_int, [_p, _i32, _p, _p, _p, _p]),
    "cvae_preprocess_u8": (C.c_int, [_p, _i32, _p, _p, _p]),
    "cvae_diff_grey": (C.c_int, [_p, _i32, _p, _p, _p, _p]),
    "cvae_probe_config": (C.c_int, [_p, C.c_uint32]),
    "cvae_probe_read": (C.c_int, [_p, _i32, C.POINTER(C.c_float), _i32]),
    "cvae_op_scratch_floats": (_i64, [_p, _i32]),
    "cvae_op_bn_partial_floats": (_i64, [_p, _i32, _i32]),
    "cvae_op_msssim_ws_floats": (_i64, [_p, _i32]),
    "cvae_op_conv_fwd": (C.c_int, [_p, _i32, _i32] + [_p] * 7),
    "cvae_op_conv_dgrad": (C.c_int, [_p, _i32, _i32] + [_p] * 6),
    "cvae_op_conv_wgrad": (C.c_int, [_p, _i32, _i32] + [_p] * 6),
    "cvae_op_d4_bwd": (C.c_int, [_p, _i32] + [_p] * 10),
    "cvae_op_bn_pool_act_fwd": (C.c_int, [_p, _i32, _i32] + [_p] * 9 + [_i32, _p]),
    "cvae_op_bn_pool_act_bwd": (C.c_int, [_p, _i32, _i32] + [_p] * 11),
    "cvae_op_msssim": (C.c_int, [_p, _i32] + [_p] * 6),
}
EXPORTS = tuple(_SIGS)


def load():
    """dlopen the library (no GPU needed) and bind every symbol include/cvae.h declares."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CvaeError(f"{LIB_PATH} not found: run __graft_entry__.build() / make -C critic-vae_amd/csrc "
                        "(the HIP library is required; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = missing export
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
        f"need a contiguous fp32 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Handle:
    """Opaque library handle + the flat-parameter layout it reports."""

    PRECISIONS = {"f32": 0, "bf16": 1, "bf16x9": 2, "bf16x6": 3}

    def __init__(self, width=64, max_batch=256, overlap_wgrad=None, precision="f32"):
        """precision "f32": every contraction on the exact-fp32 MFMA (the 1e-4 parity path).
        "bf16": forward and input-gradient convs of E2..E4 / D0 on the bf16 MFMA (fp32 accumulate,
        fp32 tensors in HBM; BASELINE.json configs 3-5).
        "bf16x9": fp32 emulation — forward / input-gradient convs on the bf16 MFMA with both operands split
        exactly into three bf16 parts (nine exact partial products per fp32 product, fp32 accumulate);
        weight gradients on the fp32 MFMA.  Meets the same 1e-4 parity bar as "f32".
        "bf16x6": as "bf16x9" with the six leading partial products (drops <= 3*2^-24 of each product)."""
        self.lib = load()
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        self.precision = precision
        if overlap_wgrad is None:      # unset: the experiment switch CVAE_OVERLAP_WGRAD=1 decides; an explicit
            overlap_wgrad = os.environ.get("CVAE_OVERLAP_WGRAD") == "1"       # argument always wins (tests compare True with False)
        self.overlap_wgrad = bool(overlap_wgrad)                               # the effective value, for bench lines and tests
        cfg = _Config(width, max_batch, int(self.overlap_wgrad), self.PRECISIONS[precision])
        h = _p()
        rc = self.lib.cvae_create(C.byref(cfg), C.byref(h))
        self._check(rc)
        self.h = h
        self.width, self.max_batch = width, max_batch
        self.param_total = self.lib.cvae_param_total(h)
        self.layout = {}
        for i in range(self.lib.cvae_param_count(h)):
            self.layout[self.lib.cvae_param_name(h, i).decode()] = (
                self.lib.cvae_param_offset(h, i), self.lib.cvae_param_numel(h, i))
        self.bn_state_floats = self.lib.cvae_bn_state_floats(h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.cvae_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CvaeError(f"libcvae_hip error {rc}: {self.lib.cvae_last_error().decode()}")

    def workspace_bytes(self, batch):
        return self.lib.cvae_workspace_bytes(self.h, batch)

    def ws_view(self, ws, batch, name, numel):
        off = self.lib.cvae_ws_offset(self.h, batch, name.encode())
        if off < 0:
            raise KeyError(name)
        return ws[off:off + numel]

    # ---- the hot path ----
    def forward(self, B, x, pred, eps, params, bn_state, mu, logvar, recon, ws, train=True):
        self._check(self.lib.cvae_forward(self.h, B, _ptr(x), _ptr(pred), _ptr(eps), _ptr(params), _ptr(bn_state),
                                          _ptr(mu), _ptr(logvar), _ptr(recon), _ptr(ws), int(train), _stream()))

    def decode(self, B, zcat, params, recon, ws):
        self._check(self.lib.cvae_decode(self.h, B, _ptr(zcat), _ptr(params), _ptr(recon), _ptr(ws), _stream()))

    def loss(self, B, x, mu, logvar, recon, ws, scalars, d_recon=None, d_mu=None, d_logvar=None):
        self._check(self.lib.cvae_loss(self.h, B, _ptr(x), _ptr(mu), _ptr(logvar), _ptr(recon), _ptr(ws),
                                       _ptr(scalars), _ptr(d_recon), _ptr(d_mu), _ptr(d_logvar), _stream()))

    def backward(self, B, x, pred, eps, params, logvar, recon, d_recon, d_mu, d_logvar, ws, grads, zero_padding=False):
        """zero_padding: `grads` is uninitialised memory — also write 0 into the alignment gaps (phase bit 3)."""
        if zero_padding:
            self._check(self.lib.cvae_backward_phases(self.h, B, _ptr(x), _ptr(pred), _ptr(eps), _ptr(params),
                                                      _ptr(logvar), _ptr(recon), _ptr(d_recon), _ptr(d_mu),
                                                      _ptr(d_logvar), _ptr(ws), _ptr(grads), 15, _stream()))
            return
        self._check(self.lib.cvae_backward(self.h, B, _ptr(x), _ptr(pred), _ptr(eps), _ptr(params), _ptr(logvar),
                                           _ptr(recon), _ptr(d_recon), _ptr(d_mu), _ptr(d_logvar), _ptr(ws),
                                           _ptr(grads), _stream()))

    def scale_loss_grads(self, B, g, d_recon, d_mu, d_logvar, out_recon, out_mu, out_logvar):
        self._check(self.lib.cvae_scale_loss_grads(self.h, B, _ptr(g), _ptr(d_recon), _ptr(d_mu), _ptr(d_logvar),
                                                   _ptr(out_recon), _ptr(out_mu), _ptr(out_logvar), _stream()))

    def grads_to_bf16(self, grads, out_bf16):
        """fp32 gradient range -> caller's bf16 buffer of the same length (transport of the optional bf16 all-reduce)."""
        assert out_bf16.is_cuda and out_bf16.dtype == torch.bfloat16 and out_bf16.is_contiguous() and out_bf16.numel() == grads.numel()
        self._check(self.lib.cvae_grads_to_bf16(self.h, _ptr(grads), out_bf16.data_ptr(), grads.numel(), _stream()))

    def grads_from_bf16(self, in_bf16, grads):
        assert in_bf16.is_cuda and in_bf16.dtype == torch.bfloat16 and in_bf16.is_contiguous() and in_bf16.numel() == grads.numel()
        self._check(self.lib.cvae_grads_from_bf16(self.h, in_bf16.data_ptr(), _ptr(grads), grads.numel(), _stream()))

    def backward_phase(self, phase, B, x, pred, eps, params, logvar, recon, d_recon, d_mu, d_logvar, ws, grads):
        """Phase 0..2 of the backward (decoder | fc + encoder block 3 | encoder blocks 2..0), in order."""
        self._check(self.lib.cvae_backward_phases(self.h, B, _ptr(x), _ptr(pred), _ptr(eps), _ptr(params), _ptr(logvar),
                                                  _ptr(recon), _ptr(d_recon), _ptr(d_mu), _ptr(d_logvar), _ptr(ws),
                                                  _ptr(grads), 1 << phase, _stream()))

    def grad_bucket(self, phase):
        """(offset, numel) of the flat-gradient range that backward phase `phase` completes."""
        off, n = C.c_int64(), C.c_int64()
        self._check(self.lib.cvae_grad_bucket(self.h, phase, C.byref(off), C.byref(n)))
        return off.value, n.value

    def adam_step(self, params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        self._check(self.lib.cvae_adam_step(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(),
                                            step, lr, b1, b2, eps, grad_scale, _stream()))

    # ---- critic + input pipeline ----
    def critic_forward(self, B, x, critic_params, pred):
        self._check(self.lib.cvae_critic_forward(self.h, B, _ptr(x), _ptr(critic_params), _ptr(pred), _stream()))

    def preprocess_u8(self, B, frames_u8, x):
        assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
        self._check(self.lib.cvae_preprocess_u8(self.h, B, frames_u8.data_ptr(), _ptr(x), _stream()))

    def diff_grey(self, B, recon_one, recon_zero, diff):
        self._check(self.lib.cvae_diff_grey(self.h, B, _ptr(recon_one), _ptr(recon_zero), _ptr(diff), _stream()))

    # ---- in-step kernel probe (bench.py roofline) ----
    def probe_config(self, ids):
        mask = 0
        for i in ids:
            mask |= 1 << i
        self._check(self.lib.cvae_probe_config(self.h, mask))

    def probe_read(self, pid, cap=128):
        buf = (C.c_float * cap)()
        n = self.lib.cvae_probe_read(self.h, pid, buf, cap)
        return [buf[i] for i in range(n)]

    # ---- per-op entry points (tests, roofline probe) ----
    def op_scratch_floats(self, B):
        return self.lib.cvae_op_scratch_floats(self.h, B)

    def op_bn_partial_floats(self, layer, B):
        return self.lib.cvae_op_bn_partial_floats(self.h, layer, B)

    def op_msssim_ws_floats(self, B):
        return self.lib.cvae_op_msssim_ws_floats(self.h, B)

    def op_conv_fwd(self, layer, B, inp, w, bias, out, bn_partials=None, scratch=None):
        self._check(self.lib.cvae_op_conv_fwd(self.h, layer, B, _ptr(inp), _ptr(w), _ptr(bias), _ptr(out),
                                              _ptr(bn_partials), _ptr(scratch), _stream()))

    def op_conv_dgrad(self, layer, B, dout, w, mask_src, din, scratch=None):
        self._check(self.lib.cvae_op_conv_dgrad(self.h, layer, B, _ptr(dout), _ptr(w), _ptr(mask_src), _ptr(din),
                                                _ptr(scratch), _stream()))

    def op_conv_wgrad(self, layer, B, inp, dout, dw, dbias, scratch):
        self._check(self.lib.cvae_op_conv_wgrad(self.h, layer, B, _ptr(inp), _ptr(dout), _ptr(dw), _ptr(dbias),
                                                _ptr(scratch), _stream()))

    def op_d4_bwd(self, B, o3, d_recon, recon, w, dout, d_o3, dw, db, scratch):
        self._check(self.lib.cvae_op_d4_bwd(self.h, B, _ptr(o3), _ptr(d_recon), _ptr(recon), _ptr(w), _ptr(dout),
                                            _ptr(d_o3), _ptr(dw), _ptr(db), _ptr(scratch), _stream()))

    def op_bn_pool_act_fwd(self, layer, B, y, bn_partials, gamma, beta, run_mean, run_var, coef, a, scratch, train=True):
        self._check(self.lib.cvae_op_bn_pool_act_fwd(self.h, layer, B, _ptr(y), _ptr(bn_partials), _ptr(gamma),
                                                     _ptr(beta), _ptr(run_mean), _ptr(run_var), _ptr(coef), _ptr(a),
                                                     _ptr(scratch), int(train), _stream()))

    def op_bn_pool_act_bwd(self, layer, B, y, a, da, coef, gamma, dy, dgamma, dbeta, dbias, scratch):
        self._check(self.lib.cvae_op_bn_pool_act_bwd(self.h, layer, B, _ptr(y), _ptr(a), _ptr(da), _ptr(coef),
                                                     _ptr(gamma), _ptr(dy), _ptr(dgamma), _ptr(dbeta), _ptr(dbias),
                                                     _ptr(scratch), _stream()))

    def op_msssim(self, B, img1, img2, ws, scalars, d_img1=None):
        self._check(self.lib.cvae_op_msssim(self.h, B, _ptr(img1), _ptr(img2), _ptr(ws), _ptr(scalars),
                                            _ptr(d_img1), _stream()))
