"""Regenerate the selfnorm_* fixtures from the REFERENCE'S OWN classes (run in the build container only; nothing of the
reference travels -- the fixtures are inputs and expected outputs).

inf.layers.selfnorm does not import as it stands here: `import wandb` (not installed) and an import-time JIT build of
inf/utils/convbackward/conv2d_backward.cpp into the read-only tree, whose two ATen entry points
(at::cudnn_convolution_backward_weight / _input, conv2d_backward.cpp:18,43) no longer exist in torch 2.x -- ordinary Python
errors.  This script puts two stand-ins into sys.modules before the import:
  * `wandb`: an empty module (only used for logging);
  * `inf.utils.convbackward`: `conv2d_backward.backward_weight / backward_input` with the published semantics of those two
    cuDNN calls, i.e. torch.nn.grad.conv2d_weight / conv2d_input.
Everything else -- SelfNormConvFunc.forward/backward (selfnorm.py:39-90), _compute_weight_multiple (:24-32), flip_kernel
(:35), SelfNormConv.forward / add_recon_grad (:155-229) -- is the reference's code, executed in fp64 on the CPU.

    python tests/golden/make_golden_selfnorm_ref.py          # rewrites tests/golden/selfnorm_*.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

sys.modules["wandb"] = types.ModuleType("wandb")
cb = types.ModuleType("inf.utils.convbackward")


class _Conv2dBackward:
    """conv2d_backward.cpp:7-54 with torch.nn.grad (groups, benchmark, deterministic as in the cuDNN signature)"""

    @staticmethod
    def backward_weight(wshape, grad_output, inp, padding, stride, dilation, groups, benchmark, deterministic):
        return torch.nn.grad.conv2d_weight(inp, tuple(wshape), grad_output, tuple(stride), tuple(padding), tuple(dilation), groups)

    @staticmethod
    def backward_input(input_size, grad_output, weight, padding, stride, dilation, groups, benchmark, deterministic):
        return torch.nn.grad.conv2d_input(tuple(input_size), weight, grad_output, tuple(stride), tuple(padding), tuple(dilation), groups)


cb.conv2d_backward = _Conv2dBackward
cb.conv_bias_map = lambda b, shape: b.unsqueeze(-1).unsqueeze(-1).unsqueeze(0).repeat(shape[0], 1, shape[2], shape[3])
sys.modules["inf.utils.convbackward"] = cb

from inf.layers import selfnorm as ref  # noqa: E402  (the reference module itself)


def selfnorm_case(name, B, C, H, W, K, pad, seed, bias=True):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen, dtype=torch.float64)
    Wt = torch.nn.init.dirac_(torch.empty(C, C, K, K)).double() + 0.05 * torch.randn(C, C, K, K, generator=gen, dtype=torch.float64)
    R = torch.flip(Wt, (2, 3)).permute(1, 0, 2, 3).clone() + 0.02 * torch.randn(C, C, K, K, generator=gen, dtype=torch.float64)
    bw = 0.1 * torch.randn(C, generator=gen, dtype=torch.float64) if bias else None
    p = (pad, pad)
    # ---- SelfNormConvFunc (selfnorm.py:39-95) through autograd
    xr = x.clone().requires_grad_(True)
    Wp = Wt.clone().requires_grad_(True)
    Rp = R.clone().requires_grad_(True)
    bp = bw.clone().requires_grad_(True) if bias else None
    z = ref.selfnorm_conv_2d(xr, Wp, bp, Rp, (1, 1), p, (1, 1), 1)
    gz = torch.randn(z.shape, generator=gen, dtype=torch.float64)
    z.backward(gz)
    multiple = ref._compute_weight_multiple(Wt.shape, z.detach(), x, torch.Size(p), torch.Size((1, 1)), torch.Size((1, 1)), 1, False, False)
    out = dict(x=x.numpy(), w=Wt.numpy(), r=R.numpy(), gz=gz.numpy(), pad=pad, z=z.detach().numpy(), dx=xr.grad.numpy(),
               dw_fwd=Wp.grad.numpy(), dw_inv=Rp.grad.numpy(), multiple=multiple.numpy())
    if bias:
        out.update(bias=bw.numpy(), dbias=bp.grad.numpy())
    # ---- SelfNormConv.add_recon_grad (selfnorm.py:187-229): recon_loss_weight 1, both variants, on the layer itself
    for sym in (False, True):
        layer = ref.SelfNormConv(C, C, (K, K), bias=bias, stride=1, padding=pad, sym_recon_grad=sym, recon_loss_weight=1.0).double()
        with torch.no_grad():
            layer.weight_fwd.copy_(Wt)
            layer.weight_inv.copy_(R)
            if bias:
                layer.bias_fwd.copy_(bw)
        layer(x)  # (stores the input)
        loss = layer.add_recon_grad()
        tag = "sym" if sym else "asym"
        out["recon_loss_" + tag] = float(loss)
        out["recon_dw_" + tag] = layer.weight_fwd.grad.numpy()
        out["recon_dr_" + tag] = layer.weight_inv.grad.numpy()
    out["source"] = "reference classes: inf.layers.selfnorm.SelfNormConvFunc / SelfNormConv"
    path = os.path.join(HERE, name + ".npz")
    old = None
    if os.path.exists(path):
        with np.load(path) as d:
            old = {k: d[k] for k in d.files}
    np.savez_compressed(path, **out)
    if old is not None:  # how far the earlier restatement-generated fixture was from the reference class
        worst = max(float(np.max(np.abs(old[k] - out[k]))) for k in old if k in out and k not in ("pad", "source"))
        print("wrote %s (max |difference| to the earlier fixture: %.3g)" % (path, worst))
    else:
        print("wrote", path)


if __name__ == "__main__":
    torch.set_num_threads(4)
    selfnorm_case("selfnorm_b3c4_8x8_k3_p1", 3, 4, 8, 8, 3, 1, seed=20)
    selfnorm_case("selfnorm_b2c8_6x6_k3_p1_nobias", 2, 8, 6, 6, 3, 1, seed=21, bias=False)
    selfnorm_case("selfnorm_b2c6_5x5_k1_p0", 2, 6, 5, 5, 1, 0, seed=22)
