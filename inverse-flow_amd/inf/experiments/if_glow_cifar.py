"""The 32x32x3 Glow with inverse-flow layers (BASELINE.json configs[3]: if_glow_cifar; configs[4]'s ImageNet-32 model is the
same builder, if_glow_imagenet32.py here) built from this package's layers: the model builder of the reference experiments,
inf/experiments/if_glow_cifar.py:28-100 (= if_glow_imagenet32.py:55-127 = if_multiGPU_imagenet32.py:176-248), with the
config defaults of if_glow_cifar.py:108-190.  Per block a Squeeze, then block_size x [ActNorm, inverse-flow layer(s),
activation, Coupling] -- no activation behind the very last step -- a SplitPrior between blocks, a standard-normal base.
Same module order, so a reference state_dict loads as is.

As in the reference, the builder's spline is the shared-weight one with 10 bins and a tail bound of 20 whatever `n_bins` /
`tail_bound` say (its `activations` table is fixed, if_glow_cifar.py:23-26): the two arguments are accepted and ignored
unless `spline_from_args=True`.  Only the switches the inverse-flow experiments use are kept (the SelfNormConv / FInC
variants of the reference builder are other models)."""
from inf.layers.actnorm import ActNorm
from inf.layers.activations import SmoothLeakyRelu, SplineActivation
from inf.layers.coupling import Coupling
from inf.layers.distributions.uniform import UniformDistribution
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import inv_flow_no_pad, inv_flow_with_pad
from inf.layers.preprocess import Dequantization, LogitTransform, Normalization
from inf.layers.splitprior import SplitPrior
from inf.layers.squeeze import Squeeze
from inf.train.losses import NegativeGaussianLoss

# inf/experiments/if_glow_cifar.py:108-190 (BASELINE configs[3] shards its batch of 256 over eight GPUs: 32 per rank)
DEFAULT_CONFIG = dict(num_blocks=2, block_size=16, coupling_width=128, batch_size=140, actnorm=False, split_prior=True,
                      activation="Spline", n_bins=7, tail_bound=10, inv_flow=False, inv_conv=False, inv_conv_no_pad=True,
                      if_kernel_size=2, lr=1e-4, grad_clip_norm=True, grad_clip=0.01, modified_grad=False, add_recon_grad=False)


def create_model(inv_conv=False, inv_flow=True, inv_conv_no_pad=False, num_blocks=3, block_size=32, coupling_width=512,
                 if_kernel_size=3, tail_bound=30, n_bins=10, activation="Spline", actnorm=True, split_prior=True,
                 image_size=(3, 32, 32), dequantize=True, split_width=512, spline_from_args=False, reference_init=False):
    alpha = 1e-6
    bins, tail = (n_bins, tail_bound) if spline_from_args else (10, 20)
    acts = {"SLR": lambda size: SmoothLeakyRelu(alpha=0.3),
            "Spline": lambda size: SplineActivation(size, n_bins=bins, tail_bound=tail, individual_weights=False)}
    layers = [Dequantization(UniformDistribution(size=image_size))] if dequantize else []
    layers += [Normalization(translation=0, scale=256), Normalization(translation=-alpha, scale=1 / (1 - 2 * alpha)),
               LogitTransform()]
    size = tuple(image_size)
    for block in range(num_blocks):
        layers.append(Squeeze())
        size = (size[0] * 4, size[1] // 2, size[2] // 2)
        for k in range(block_size):
            if actnorm:
                layers.append(ActNorm(size[0]))
            if inv_conv:
                layers.append(inv_flow_with_pad(size[0], size[0], (3, 3), order="TL", reference_init=reference_init))
            if inv_flow:
                layers.append(inv_flow_with_pad(size[0], size[0], (if_kernel_size, if_kernel_size), order="TL", reference_init=reference_init))
            if inv_conv_no_pad:
                layers.append(inv_flow_no_pad(size[0], size[0], (3, 3), reference_init=reference_init))
            if activation in acts and not (block == num_blocks - 1 and k == block_size - 1):
                layers.append(acts[activation](size))
            layers.append(Coupling(size, width=coupling_width))
        if split_prior and block < num_blocks - 1:
            layers.append(SplitPrior(size, NegativeGaussianLoss, width=split_width))  # (the reference keeps the default 512)
            size = (size[0] // 2, size[1], size[2])
    return FlowSequential(NegativeGaussianLoss(size=size), *layers)
