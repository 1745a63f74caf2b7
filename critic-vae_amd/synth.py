"""Deterministic counter-based generator for weights, frames, critic values and noise.

Every value is a pure function of (seed, tensor name, flat index in the REFERENCE layout),
so goldens can be regenerated anywhere (this container, the GPU box) without the reference,
without shipping weights and without depending on torch's RNG streams.  numpy only.

Reference shapes / init distributions follow PyTorch defaults as used by vae_nets.py:68-99,
116-137 (kaiming-uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for conv/linear weight and bias;
BatchNorm gamma=1, beta=0).
"""
import zlib

import numpy as np

from . import params as P

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _key(seed, name):
    s = np.uint64((int(seed) * 0x100000001B3 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF)
    return _mix64(s)


def _bits(seed, name, n, lane=0):
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix64(_key(seed, name) ^ _mix64(idx * np.uint64(2) + np.uint64(lane)))


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    """float32 U[lo, hi) with 24-bit resolution, indexed by flat reference index."""
    n = int(np.prod(shape))
    u = (_bits(seed, name, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape):
    """float32 N(0,1) via Box-Muller on two independent 24-bit uniforms."""
    n = int(np.prod(shape))
    u1 = ((_bits(seed, name, n, 0) >> np.uint64(40)).astype(np.float64) + 1.0) / float(1 << 24)
    u2 = (_bits(seed, name, n, 1) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def reference_param_shapes(width=P.w, dims=P.dims):
    """Ordered {name: shape} in the reference's state_dict naming and layouts
    (vae_nets.py:68-99 encoder, :116-137 decoder; conv OIHW, linear [out,in])."""
    bott = P.bottleneck_for(width)
    shapes = {}
    enc_ch = (P.ch,) + tuple(dims)
    for i in range(4):
        ci, co = enc_ch[i], enc_ch[i + 1]
        shapes[f"encoder.model.{4 * i}.weight"] = (co, ci, P.k, P.k)
        shapes[f"encoder.model.{4 * i}.bias"] = (co,)
        shapes[f"encoder.model.{4 * i + 1}.weight"] = (co,)
        shapes[f"encoder.model.{4 * i + 1}.bias"] = (co,)
    for nm in ("fc_mu", "fc_var"):
        shapes[f"encoder.{nm}.weight"] = (P.latent_dim, bott)
        shapes[f"encoder.{nm}.bias"] = (P.latent_dim,)
    dec_ch = (dims[3], dims[2], dims[1], dims[0], dims[0], P.ch)
    for i in range(5):
        ci, co = dec_ch[i], dec_ch[i + 1]
        shapes[f"decoder.model.{3 * i}.weight"] = (co, ci, P.k, P.k)
        shapes[f"decoder.model.{3 * i}.bias"] = (co,)
    shapes["decoder.decoder_input.weight"] = (bott, P.latent_dim + 1)
    shapes["decoder.decoder_input.bias"] = (bott,)
    return shapes


def _fan_in(name, shape, shapes):
    if name.endswith(".weight"):
        return int(np.prod(shape[1:]))
    return int(np.prod(shapes[name[:-5] + ".weight"][1:]))      # bias: fan_in of its weight


def make_params(seed=0, width=P.w, dims=P.dims):
    """{name: float32 ndarray} in reference layout, PyTorch-default init distribution."""
    shapes = reference_param_shapes(width, dims)
    bn_names = {f"encoder.model.{4 * i + 1}" for i in range(4)}
    out = {}
    for name, shape in shapes.items():
        mod = name.rsplit(".", 1)[0]
        if mod in bn_names:
            out[name] = (np.ones if name.endswith("weight") else np.zeros)(shape, np.float32)
            continue
        bound = 1.0 / np.sqrt(_fan_in(name, shape, shapes))
        out[name] = uniform(seed, name, shape, -bound, bound)
    return out


def make_batch(seed, step, batch, width=P.w, first_index=0):
    """Frames x (B,3,w,w) U[0,1), critic values pred (B,1) U[0,1), noise eps (B,32) N(0,1).

    Row b of every tensor depends only on (seed, step, first_index + b), so a data-parallel
    rank can generate exactly its shard of the global batch."""
    per = P.ch * width * width
    rows = np.arange(first_index, first_index + batch)
    x = _rows(seed, f"x/{step}", rows, per).reshape(batch, P.ch, width, width)
    pred = _rows(seed, f"pred/{step}", rows, 1)
    eps = _rows_normal(seed, f"eps/{step}", rows, P.latent_dim)
    return x.astype(np.float32), pred.astype(np.float32), eps.astype(np.float32)


def _rows(seed, name, rows, per):
    return np.stack([uniform(seed, f"{name}/{r}", (per,)) for r in rows])


def _rows_normal(seed, name, rows, per):
    return np.stack([normal(seed, f"{name}/{r}", (per,)) for r in rows])


CRITIC_SHAPES = (("features.0.weight", (8, 3, 3, 3)), ("features.0.bias", (8,)),
                 ("features.3.weight", (8, 8, 3, 3)), ("features.3.bias", (8,)),
                 ("features.6.weight", (8, 8, 3, 3)), ("features.6.bias", (8,)),
                 ("features.10.weight", (16, 8, 3, 3)), ("features.10.bias", (16,)),
                 ("features.14.weight", (32, 16, 4, 4)), ("features.14.bias", (32,)),
                 ("crit.1.weight", (32, 32)), ("crit.1.bias", (32,)),
                 ("crit.4.weight", (1, 32)), ("crit.4.bias", (1,)))


def make_critic_params(seed=0):
    """Generator-defined critic weights (critic_net.py:15-41 shapes, PyTorch-default init bounds)."""
    shapes = dict(CRITIC_SHAPES)
    out = {}
    for name, shape in CRITIC_SHAPES:
        wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shape
        bound = 1.0 / np.sqrt(int(np.prod(wshape[1:])))
        out[name] = uniform(seed, "critic." + name, shape, -bound, bound)
    return out
