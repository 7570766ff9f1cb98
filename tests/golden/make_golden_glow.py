"""Golden vectors of the Glow-step neighbours from the REFERENCE's own layers on the CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_glow.py

Imported from the reference (PYTHONPATH=/root/reference, nothing is copied): inf.layers.actnorm.ActNorm,
inf.layers.squeeze.{space_to_depth, depth_to_space}, inf.layers.coupling.Coupling.  Gradients come from
torch.autograd through the reference's forward.  The fixtures are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from inf.layers.actnorm import ActNorm  # noqa: E402
from inf.layers.coupling import Coupling  # noqa: E402
from inf.layers.squeeze import depth_to_space, space_to_depth  # noqa: E402


def n(t):
    return t.detach().numpy().copy()


def actnorm_case(name, B, C, H, W, seed, init_from_data):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3).requires_grad_(True)
    layer = ActNorm(C)
    if not init_from_data:
        with torch.no_grad():
            layer.translation.copy_(torch.randn(C, generator=g))
            layer.log_scale.copy_(torch.randn(C, generator=g) * 0.5)
            layer.initialized.fill_(1)
    y, ld = layer(x)  # (initialises from the data when not initialised: actnorm.py:21-28)
    gy = torch.randn(B, C, H, W, generator=g)
    gld = torch.randn(B, generator=g)
    (y * gy).sum().add((ld * gld).sum()).backward()
    xr = layer.reverse(y.detach())
    np.savez(os.path.join(HERE, name), x=n(x), translation=n(layer.translation), log_scale=n(layer.log_scale), y=n(y),
             logdet=n(ld), gy=n(gy), gld=n(gld), gx=n(x.grad), g_translation=n(layer.translation.grad),
             g_log_scale=n(layer.log_scale.grad), x_rev=n(xr), init_from_data=np.int32(init_from_data))


def squeeze_case(name, B, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    y = space_to_depth(x)
    np.savez(os.path.join(HERE, name), x=n(x), y=n(y), x_back=n(depth_to_space(y)))


def coupling_case(name, B, C, H, W, width, seed):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    layer = Coupling((C, H, W), width=width)
    with torch.no_grad():  # (Conv2dZero starts at zero: give the conditioner something to say)
        for p in layer.net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.15)
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    h = layer.net(x[:, :C // 2])
    h.retain_grad()
    # the layer's own forward recomputes h; feed the same values through its expressions to expose dL/dh
    h_s, t = h[:, ::2], h[:, 1::2]
    log_s = 2. * torch.tanh(h_s / 2.)
    y_l, ld_l = layer(x)
    y = torch.cat([x[:, :C // 2], x[:, C // 2:] * torch.exp(log_s) + t], dim=1)
    ld = log_s.flatten(start_dim=1).sum(-1)
    assert torch.equal(y, y_l) and torch.equal(ld, ld_l)
    gy = torch.randn(B, C, H, W, generator=g)
    gld = torch.randn(B, generator=g)
    (y * gy).sum().add((ld * gld).sum()).backward()
    xr = layer.reverse(y.detach())
    sd = {k.replace(".", "__"): n(v) for k, v in layer.state_dict().items()}
    np.savez(os.path.join(HERE, name), x=n(x), h=n(h), y=n(y), logdet=n(ld), gy=n(gy), gld=n(gld), gx_total=n(x.grad),
             gh=n(h.grad), x_rev=n(xr), width=np.int32(width), **sd)


# ---- activations (appended: SmoothLeakyRelu, SplineActivation with shared weights) ------------------------------------
def activation_cases(only_individual=False):
    from inf.layers.activations import SmoothLeakyRelu, SplineActivation

    def run(name, layer, x, extra):
        x = x.clone().requires_grad_(True)
        y, ld = layer(x)
        g = torch.Generator().manual_seed(99)
        gy = torch.randn(x.shape, generator=g)
        gld = torch.randn(x.shape[0], generator=g)
        (y * gy).sum().add((ld * gld).sum()).backward()
        with torch.no_grad():
            xr = layer.reverse(y.detach())
        d = dict(x=n(x), y=n(y), logdet=n(ld), gy=n(gy), gld=n(gld), gx=n(x.grad), x_rev=n(xr))
        for k, p in layer.named_parameters():
            d["p_" + k] = n(p)
            d["g_" + k] = n(p.grad)
        d.update(extra)
        np.savez(os.path.join(HERE, name), **d)

    # SplineActivation(individual_weights=True): one set of knots per element (the MNIST Glow's activation); its own
    # generator, so that `--individual` adds these files without rewriting the others
    gi = torch.Generator().manual_seed(12)
    for name, shape, nb, tb, scale in [("splinepe_b5c4_6x5_n5.npz", (5, 4, 6, 5), 5, 10.0, 6.0),
                                       ("splinepe_b9c3_4x4_n8_tb3.npz", (9, 3, 4, 4), 8, 3.0, 2.5),
                                       ("splinepe_b2c8_7x7_n5_tb20.npz", (2, 8, 7, 7), 5, 20.0, 9.0)]:
        layer = SplineActivation(shape[1:], n_bins=nb, tail_bound=tb, individual_weights=True)
        with torch.no_grad():
            for p in layer.parameters():
                p.copy_(torch.randn(p.shape, generator=gi) * 0.8)
        x = torch.randn(shape, generator=gi) * scale
        x[0, 0, 0, 0], x[0, 0, 0, 1] = tb, -tb
        run(name, layer, x, dict(n_bins=np.int32(nb), tail_bound=np.float32(tb)))
    if only_individual:
        return
    g = torch.Generator().manual_seed(11)
    run("slr_b3c4_6x5.npz", SmoothLeakyRelu(0.3), torch.randn(3, 4, 6, 5, generator=g) * 3, dict(alpha=np.float32(0.3)))
    run("slr_b2c6_8x8_a01.npz", SmoothLeakyRelu(0.1), torch.randn(2, 6, 8, 8, generator=g) * 2, dict(alpha=np.float32(0.1)))
    for name, shape, nb, tb, scale in [("spline_b3c4_6x5_n5.npz", (3, 4, 6, 5), 5, 10.0, 6.0),
                                       ("spline_b2c6_8x8_n5_tb3.npz", (2, 6, 8, 8), 5, 3.0, 2.5),
                                       ("spline_b2c3_4x4_n8.npz", (2, 3, 4, 4), 8, 5.0, 3.0)]:
        layer = SplineActivation(shape[1:], n_bins=nb, tail_bound=tb)
        with torch.no_grad():
            for p in layer.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.8)
        x = torch.randn(shape, generator=g) * scale  # (some elements beyond the tail bound: the linear tails)
        x[0, 0, 0, 0], x[0, 0, 0, 1] = tb, -tb       # the interval's end points are inside
        run(name, layer, x, dict(n_bins=np.int32(nb), tail_bound=np.float32(tb)))


if __name__ == "__main__":
    if "--individual" in sys.argv:
        activation_cases(only_individual=True)
        sys.exit(0)
    actnorm_case("actnorm_b3c6_8x8.npz", 3, 6, 8, 8, 1, False)
    actnorm_case("actnorm_b4c5_7x5_datainit.npz", 4, 5, 7, 5, 2, True)
    actnorm_case("actnorm_b2c12_16x16.npz", 2, 12, 16, 16, 3, False)
    squeeze_case("squeeze_b2c3_8x12.npz", 2, 3, 8, 12, 4)
    squeeze_case("squeeze_b1c2_6x6.npz", 1, 2, 6, 6, 5)
    coupling_case("coupling_b2c8_6x6_w16.npz", 2, 8, 6, 6, 16, 6)
    coupling_case("coupling_b3c12_8x8_w24.npz", 3, 12, 8, 8, 24, 7)
    coupling_case("coupling_b2c6_5x7_w8.npz", 2, 6, 5, 7, 8, 8)
    activation_cases()
    print("wrote glow-step fixtures to", HERE)
