"""The ImageNet-32 Glow with inverse-flow layers (BASELINE.json configs[4]): the reference's builder for it
(inf/experiments/if_glow_imagenet32.py:55-127, inf/if_multiGPU_imagenet32.py:176-248) is the 32x32x3 builder of
if_glow_cifar.py; what differs is the configuration.  DEFAULT_CONFIG is the multi-GPU script's
(if_multiGPU_imagenet32.py:284-345: three blocks of 48 steps, no ActNorm, 3x3 inverse-flow layers without padding order,
batch 100 on nn.DataParallel -- here one process per GPU and a shard of the batch each, data_parallel.py);
SINGLE_GPU_CONFIG the single-GPU experiment's (if_glow_imagenet32.py:141-200)."""
from inf.experiments.if_glow_cifar import create_model  # noqa: F401

DEFAULT_CONFIG = dict(num_blocks=3, block_size=48, coupling_width=256, batch_size=100, actnorm=False, split_prior=True,
                      activation="Spline", n_bins=7, tail_bound=10, inv_flow=False, inv_conv=False, inv_conv_no_pad=True,
                      if_kernel_size=2, lr=1e-3, grad_clip_norm=0.06, grad_clip=0.01, modified_grad=False, add_recon_grad=False)
SINGLE_GPU_CONFIG = dict(DEFAULT_CONFIG, num_blocks=2, block_size=32, lr=1e-4, grad_clip_norm=True)
