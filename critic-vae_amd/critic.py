"""Host-side mirror of the reference's frozen critic (critic_net.py:5-69) for the training loop:
`preds = critic.evaluate(images)` (vae.py:50).  Inference only; parameters are kept as one flat
buffer in the reference's state_dict order (what cvae_critic_forward reads)."""
import torch
from torch import nn

from .lib import Handle

# reference state_dict keys, in order, with shapes (critic_net.py:15-41, default arguments)
CRITIC_KEYS = (("features.0.weight", (8, 3, 3, 3)), ("features.0.bias", (8,)),
               ("features.3.weight", (8, 8, 3, 3)), ("features.3.bias", (8,)),
               ("features.6.weight", (8, 8, 3, 3)), ("features.6.bias", (8,)),
               ("features.10.weight", (16, 8, 3, 3)), ("features.10.bias", (16,)),
               ("features.14.weight", (32, 16, 4, 4)), ("features.14.bias", (32,)),
               ("crit.1.weight", (32, 32)), ("crit.1.bias", (32,)),
               ("crit.4.weight", (1, 32)), ("crit.4.bias", (1,)))


class Critic(nn.Module):
    def __init__(self, width=64, handle=None):
        super().__init__()
        self.width = width
        self.handle = handle if handle is not None else Handle(width, 1)
        n = self.handle.lib.cvae_critic_param_count()
        assert n == sum(int(torch.tensor(s).prod()) for _, s in CRITIC_KEYS)
        self.register_buffer("flat", torch.zeros(n))

    def load_state_dict(self, sd, strict=True):
        """Accepts the reference's checkpoint (saved-networks/critic-*.pt, vae_utility.py:363-370)."""
        parts = []
        for k, shape in CRITIC_KEYS:
            t = sd[k]
            assert tuple(t.shape) == shape, (k, tuple(t.shape), shape)
            parts.append(t.reshape(-1).to(torch.float32))
        with torch.no_grad():
            self.flat.copy_(torch.cat(parts))

    def state_dict(self, *a, **k):
        out, off = {}, 0
        for key, shape in CRITIC_KEYS:
            n = int(torch.tensor(shape).prod())
            out[key] = self.flat[off:off + n].reshape(shape).clone()
            off += n
        return out

    def preprocess(self, X):
        """critic_net.py:60-63 / vae_utility.py:337-343: uint8 (B,H,W,3) -> float (B,3,H,W) / 255."""
        X = X.contiguous()
        out = torch.empty(X.shape[0], 3, self.width, self.width, device=X.device)
        self.handle.preprocess_u8(X.shape[0], X, out)
        return out

    def forward(self, X, collect=False):
        if collect:
            raise NotImplementedError("collect=True (embeddings) is not on the training path")
        return self.evaluate(X)

    def evaluate(self, X):
        """critic_net.py:66-69: no_grad forward, returns (B,1) in (0,1)."""
        with torch.no_grad():
            X = X.to(torch.float32).contiguous()
            pred = torch.empty(X.shape[0], 1, device=X.device)
            self.handle.critic_forward(X.shape[0], X, self.flat, pred)
        return pred
