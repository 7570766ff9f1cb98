"""The MNIST Glow with inverse-flow layers (BASELINE.json configs[2]: if_glow_mnist, L = 2 blocks of K = 16 steps) built
from this package's layers: the model builder of the reference experiment, inf/experiments/if_glow_mnist.py:33-132, with
its config defaults (:156-190): per block a Squeeze, then block_size x [ActNorm, inverse-flow layer, activation, Coupling],
a SplitPrior between blocks, a standard-normal base.  Same module order, so a reference state_dict loads as is.  Only the
switches the inverse-flow experiments use are kept (the SelfNormConv / FInC variants of the reference builder are other
models)."""
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

# inf/experiments/if_glow_mnist.py:156-190
DEFAULT_CONFIG = dict(num_blocks=2, block_size=16, coupling_width=512, batch_size=100, actnorm=True, split_prior=True,
                      activation="Spline", n_bins=5, tail_bound=20, inv_flow=False, inv_conv_no_pad=True, if_kernel_size=3,
                      lr=1e-5, grad_clip_norm=True, grad_clip=0.01, modified_grad=True, add_recon_grad=True)


def create_model(inv_flow=False, inv_conv_no_pad=True, if_kernel_size=3, coupling_width=512, num_blocks=2, block_size=16,
                 tail_bound=20, n_bins=5, actnorm=True, activation="Spline", split_prior=True, image_size=(1, 28, 28),
                 dequantize=True, split_width=512, reference_init=False):
    alpha = 1e-7
    acts = {"SLR": lambda size: SmoothLeakyRelu(alpha=0.3),
            "Spline": lambda size: SplineActivation(size, n_bins=n_bins, tail_bound=tail_bound, individual_weights=True)}
    layers = [Dequantization(UniformDistribution(size=image_size))] if dequantize else []
    layers += [Normalization(translation=0, scale=256), Normalization(translation=-alpha, scale=1 / (1 - 2 * alpha)),
               LogitTransform()]
    size = tuple(image_size)
    for block in range(num_blocks):
        layers.append(Squeeze())
        size = (size[0] * 4, size[1] // 2, size[2] // 2)
        for _ in range(block_size):
            if actnorm:
                layers.append(ActNorm(size[0]))
            if inv_flow:
                layers.append(inv_flow_with_pad(size[0], size[0], (if_kernel_size, if_kernel_size), order="TL", reference_init=reference_init))
            if inv_conv_no_pad:
                layers.append(inv_flow_no_pad(size[0], size[0], (2, 2), reference_init=reference_init))
            if activation in acts:
                layers.append(acts[activation](size))
            layers.append(Coupling(size, width=coupling_width))
        if split_prior and block < num_blocks - 1:
            layers.append(SplitPrior(size, NegativeGaussianLoss, width=split_width))  # (the reference keeps the default 512)
            size = (size[0] // 2, size[1], size[2])
    return FlowSequential(NegativeGaussianLoss(size=size), *layers)
