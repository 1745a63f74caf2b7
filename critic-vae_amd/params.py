"""Hyper-parameters of the Critic-VAE training path.

Same names and values as the reference's ``vae_parameters.py:5-22`` (module constants that
``vae.py`` star-imports).  ``bottleneck`` is derived from ``w`` so that 128x128 frames
(BASELINE.json config 5) are expressible; at w=64 it is the reference's 4096.
"""

### IMAGE DATA ###
w = 64          # frame width == height (vae_parameters.py:5)
ch = 3          # image channels (vae_parameters.py:6)

### TRAINING DATA ###
epochs = 7              # vae_parameters.py:9
batch_size = 128        # vae_parameters.py:10
lr = 0.00005            # vae_parameters.py:11
k = 5                   # conv kernel size (vae_parameters.py:12)
p = 2                   # conv padding (vae_parameters.py:13)
step = 1                # conv stride (vae_parameters.py:14)
latent_dim = 32         # vae_parameters.py:16
kld_weight = 0.001      # vae_parameters.py:17
total_images = 50000    # vae_parameters.py:19
log_n = batch_size * 30  # vae_parameters.py:21
inject_n = 6            # vae_parameters.py:22

dims = (32, 64, 128, 256)   # default channel plan (vae_nets.py:8)


def bottleneck_for(width: int) -> int:
    """256 * (w/16)^2; 4096 at w=64 (vae_parameters.py:15)."""
    return 256 * (width // 16) ** 2


bottleneck = bottleneck_for(w)

# Adam defaults used by vae.py:36 (torch.optim.Adam(lr=lr))
adam_betas = (0.9, 0.999)
adam_eps = 1e-8

# MS-SSIM constants (vae_nets.py:152-154, 171, 201-203, 219)
msssim_window = 11
msssim_sigma = 1.5
msssim_weights = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
msssim_C1 = 0.01 ** 2
msssim_C2 = 0.03 ** 2

# BatchNorm2d defaults (vae_nets.py:70)
bn_eps = 1e-5
bn_momentum = 0.1
