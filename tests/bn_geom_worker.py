"""Child process of test_gpu_bf16.py::test_big_tile_bn_partials_match_the_per_tile_kernels: one bf16 forward (train mode) at the given frame
size / batch with whatever conv kernels the environment selects (CVAE_BF16_BIG), running statistics and outputs saved to an .npz."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from critic_vae_amd import synth  # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402

W, B, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
x, pred, eps = (torch.from_numpy(v).to(dev) for v in synth.make_batch(1234, 0, B, W))
vae = VariationalAutoencoder(width=W, max_batch=B, seed=0, precision="bf16").to(dev)
vae.load_reference_params(synth.make_params(0, W))
_, mu, logvar, recon = vae(x, pred, eps=eps)
torch.cuda.synchronize()
sd = vae.encoder.state_dict()
np.savez(out, mu=mu.detach().cpu().numpy(), logvar=logvar.detach().cpu().numpy(),
         **{f"rm{bi}": sd[f"model.{bi}.running_mean"].cpu().numpy() for bi in (1, 5, 9, 13)},
         **{f"rv{bi}": sd[f"model.{bi}.running_var"].cpu().numpy() for bi in (1, 5, 9, 13)})
