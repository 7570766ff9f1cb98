"""Activation layers of the flow on the HIP library (reference: inf/layers/activations.py).

FlowActivationLayer keeps the reference's generic surface (forward = (activation, logdet), logdet from act_prime).
SmoothLeakyRelu and SplineActivation (shared weights) run 4-D CUDA inputs through one-pass kernels of libinvflow_hip:
  * SmoothLeakyRelu (activations.py:37-54): value, log-derivative sum and -- for reverse -- the reference's 100 Newton
    iterations in registers instead of 100 passes over the tensor;
  * SplineActivation (activations.py:126-217): the reference repeats its 3 n_bins - 1 parameters to three
    (B, C, H, W, n_bins) tensors and runs splines/rational_quadratic.py on them; here the knot tables (n_bins + 1
    entries each) are computed from the parameters by the same formulas on tiny tensors -- autograd sees that part --
    and the per-element spline with its derivative sums is one kernel each way (ifl_rqspline_f32 / _backward_f32).
Individual weights (one set of knots per element, the MNIST Glow): ifl_rqspline_pe_f32 / _backward_f32 build the tables in
registers.  Anything else (other dimensionalities, CPU tensors) takes the reference's torch expressions.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

import invflow_hip as H

from .flowlayer import FlowLayer

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")

MIN_BIN_WIDTH = MIN_BIN_HEIGHT = MIN_DERIVATIVE = 1e-6  # splines/rational_quadratic.py:7-9


class FlowActivationLayer(FlowLayer):
    def __init__(self):
        super().__init__()
        self._last_logdet_value = None

    def forward(self, input, context=None):
        act = self.activation(input, context)
        return act, self.logdet(input, context)

    def act_prime(self, input, context=None):
        raise NotImplementedError()

    def logdet(self, input, context=None):
        logderiv = torch.log(torch.abs(self.act_prime(input, context)))
        return logderiv.flatten(start_dim=1).sum(dim=-1)

    def reverse(self, input, context=None):
        raise NotImplementedError()


def newton_raphson_inverse(f, y, x0, context=None, n_iter=100):
    x = x0  # activations.py:27-34
    for _ in range(n_iter):
        fprime = torch.clamp(f.act_prime(x, context), min=1e-2)
        x = x - (f.activation(x, context) - y) / fprime
    return x


def _hip_ok(t):
    return t.dim() == 4 and t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16)


class _SlrFn(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, alpha):
        x = x.contiguous()
        y, ld = H.slr(x, alpha)
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        (x,) = ctx.saved_tensors
        return H.slr_backward(gy.contiguous(), None if gld is None else gld.contiguous(), x, ctx.alpha), None


class SmoothLeakyRelu(FlowActivationLayer):
    def __init__(self, alpha=0.3):
        super().__init__()
        self.alpha = alpha

    def activation(self, input, context=None):
        alpha = self.alpha
        stacked = torch.stack((torch.zeros_like(input), input))
        return alpha * input + (1 - alpha) * torch.logsumexp(stacked, dim=0)

    def act_prime(self, input, context=None):
        return self.alpha + (1 - self.alpha) * torch.sigmoid(input)

    def forward(self, input, context=None):
        if _hip_ok(input):
            return _SlrFn.apply(input, float(self.alpha))
        return super().forward(input, context)

    def reverse(self, input, context=None):
        if _hip_ok(input) and input.dtype == torch.float32 and not torch.is_grad_enabled():
            return H.slr(input.contiguous(), float(self.alpha), reverse=True)
        y, x0 = input, input
        return newton_raphson_inverse(self, y, x0, context)


def spline_tables(unnormalized_widths, unnormalized_heights, unnormalized_derivatives, tail_bound):
    """Knot tables of the spline with linear tails: cumwidths, cumheights, derivatives (n_bins + 1 entries along the last
    axis), by the formulas of splines/rational_quadratic.py:35-46,97-116.  1-D parameters (shared weights) or one set per
    element (individual weights: leading axes (1, C, H, W))."""
    nb = unnormalized_widths.shape[-1]
    constant = float(np.log(np.exp(1 - MIN_DERIVATIVE) - 1))
    ud = F.pad(unnormalized_derivatives, pad=(1, 1)) + constant
    left = bottom = -tail_bound
    right = top = tail_bound

    def knots(u, lo, hi, min_size):
        v = F.softmax(u, dim=-1)
        v = min_size + (1 - min_size * nb) * v
        cum = torch.cumsum(v, dim=-1)
        cum = F.pad(cum, pad=(1, 0), mode='constant', value=0.0)
        cum = (hi - lo) * cum + lo
        # cum[0] = lo, cum[-1] = hi without an in-place write (same values, same gradients) and without a scalar written
        # into a tensor element (a host-to-device copy: not capturable into a graph)
        idx = torch.arange(cum.shape[-1], device=cum.device)
        first, last = (idx == 0).to(cum.dtype), (idx == cum.shape[-1] - 1).to(cum.dtype)
        return cum * (1.0 - first - last) + (lo * first + hi * last)

    cw = knots(unnormalized_widths, left, right, MIN_BIN_WIDTH)
    ch = knots(unnormalized_heights, bottom, top, MIN_BIN_HEIGHT)
    dv = MIN_DERIVATIVE + F.softplus(ud)
    return cw, ch, dv


class _TablesFn(torch.autograd.Function):
    """parameters -> knot tables in one launch each way (ifl_rqspline_tables_f32 / _backward_f32) instead of the ~25
    eager kernels of spline_tables() and as many again in its autograd backward"""

    @staticmethod
    def forward(ctx, uw, uh, ud, tail_bound):
        uw, uh, ud = uw.contiguous(), uh.contiguous(), ud.contiguous()
        ctx.save_for_backward(uw, uh, ud)
        ctx.tail_bound = tail_bound
        return H.rqspline_tables(uw, uh, ud, tail_bound)

    @staticmethod
    def backward(ctx, gcw, gch, gdv):
        uw, uh, ud = ctx.saved_tensors
        z = lambda g, n: torch.zeros(n, device=uw.device) if g is None else g
        n = uw.numel() + 1
        gt = torch.stack([z(gcw, n), z(gch, n), z(gdv, n)]).float().contiguous()
        return (*H.rqspline_tables_backward(gt, uw, uh, ud, ctx.tail_bound), None)


class _SplineFn(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, cw, ch, dv, tail_bound):
        x = x.contiguous()
        y, ld = H.rqspline(x, cw, ch, dv, tail_bound)
        ctx.save_for_backward(x, cw, ch, dv)
        ctx.tail_bound = tail_bound
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, cw, ch, dv = ctx.saved_tensors
        gx, gcw, gch, gdv = H.rqspline_backward(gy.contiguous(), None if gld is None else gld.contiguous(), x, cw, ch, dv,
                                                ctx.tail_bound)
        return gx, gcw, gch, gdv, None


class _SplineParamFn(torch.autograd.Function):
    """the shared-weight spline from its parameters in one launch each way (ifl_rqspline_p_f32 / _backward_f32: the knot tables
    are computed inside the spline's launch, their gradients chained to the parameters inside the reduction's) -- what
    _TablesFn + _SplineFn do in two"""

    @staticmethod
    @_fwd32
    def forward(ctx, x, uw, uh, ud, tail_bound):
        x, uw, uh, ud = x.contiguous(), uw.contiguous(), uh.contiguous(), ud.contiguous()
        y, ld, tables = H.rqspline_p(x, uw, uh, ud, tail_bound)
        ctx.save_for_backward(x, tables, uw, uh, ud)
        ctx.tail_bound = tail_bound
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, tables, uw, uh, ud = ctx.saved_tensors
        return (*H.rqspline_p_backward(gy.contiguous(), None if gld is None else gld.contiguous(), x, tables, uw, uh, ud,
                                       ctx.tail_bound), None)


class _SplinePEFn(torch.autograd.Function):
    """the spline with per-element knots: one kernel forward (tables built in registers), one backward + a fixed-order sum
    of the parameter gradients over image groups (ifl_rqspline_pe_f32 / _backward_f32)"""

    @staticmethod
    @_fwd32
    def forward(ctx, x, uw, uh, ud, tail_bound):
        x, uw, uh, ud = x.contiguous(), uw.contiguous(), uh.contiguous(), ud.contiguous()
        y, ld = H.rqspline_pe(x, uw, uh, ud, tail_bound)
        ctx.save_for_backward(x, uw, uh, ud)
        ctx.tail_bound = tail_bound
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, uw, uh, ud = ctx.saved_tensors
        gx, guw, guh, gud = H.rqspline_pe_backward(gy.contiguous(), gld.contiguous() if gld is not None else None, x, uw, uh, ud,
                                                   ctx.tail_bound)
        return gx, guw, guh, gud, None


class SplineActivation(FlowActivationLayer):
    def __init__(self, input_size, n_bins=5, tail_bound=10., individual_weights=False):
        super().__init__()
        self.n_bins = n_bins
        self.tail_bound = tail_bound
        self.individual_weights = individual_weights
        if individual_weights:
            self.unnormalized_widths = torch.nn.Parameter(torch.randn(1, *input_size, n_bins) * 0.01)
            self.unnormalized_heights = torch.nn.Parameter(torch.randn(1, *input_size, n_bins) * 0.01)
            self.unnormalized_derivatives = torch.nn.Parameter(torch.randn(1, *input_size, n_bins - 1) * 0.01)
        else:
            self.unnormalized_widths = torch.nn.Parameter(torch.randn(n_bins) * 0.01)
            self.unnormalized_heights = torch.nn.Parameter(torch.randn(n_bins) * 0.01)
            self.unnormalized_derivatives = torch.nn.Parameter(torch.randn(n_bins - 1) * 0.01)

    def _hip(self, input):
        return _hip_ok(input) and not self.individual_weights and 1 <= self.n_bins <= 16

    def _tables(self):
        p = self.unnormalized_widths
        if (p.is_cuda and p.dtype == torch.float32 and not self.individual_weights and 2 <= self.n_bins <= 16
                and not os.environ.get("IFL_TORCH_SPLINE_TABLES")):  # (the switch is for A/B timing)
            return _TablesFn.apply(self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives,
                                   float(self.tail_bound))
        return spline_tables(self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives,
                             float(self.tail_bound))

    def forward(self, input, context=None):
        return self.activation_and_logdet(input, context)

    def _from_params(self, input):
        p = self.unnormalized_widths
        return (input.dtype == torch.float32 and p.is_cuda and p.dtype == torch.float32 and 2 <= self.n_bins <= 16
                and not os.environ.get("IFL_TORCH_SPLINE_TABLES"))

    def _hip_pe(self, input):
        return (_hip_ok(input) and self.individual_weights and 1 <= self.n_bins <= 8
                and tuple(input.shape[1:]) == tuple(self.unnormalized_widths.shape[1:-1]))

    def activation_and_logdet(self, input, context=None):
        if self._hip_pe(input):
            return _SplinePEFn.apply(input, self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives,
                                     float(self.tail_bound))
        if self._hip(input):
            if self._from_params(input):
                return _SplineParamFn.apply(input, self.unnormalized_widths, self.unnormalized_heights,
                                            self.unnormalized_derivatives, float(self.tail_bound))
            cw, ch, dv = self._tables()
            return _SplineFn.apply(input, cw, ch, dv, float(self.tail_bound))
        return _spline_torch(self, input, inverse=False)

    def reverse(self, input, context=None):
        if self._hip_pe(input) and input.dtype == torch.float32 and not torch.is_grad_enabled():
            return H.rqspline_pe(input.contiguous(), self.unnormalized_widths.contiguous(), self.unnormalized_heights.contiguous(),
                                 self.unnormalized_derivatives.contiguous(), float(self.tail_bound), inverse=True, want_logdet=False)[0]
        if self._hip(input) and input.dtype == torch.float32 and not torch.is_grad_enabled():
            if self._from_params(input):
                return H.rqspline_p(input.contiguous(), self.unnormalized_widths.contiguous(), self.unnormalized_heights.contiguous(),
                                    self.unnormalized_derivatives.contiguous(), float(self.tail_bound), inverse=True,
                                    want_logdet=False)[0]
            cw, ch, dv = self._tables()
            return H.rqspline(input.contiguous(), cw, ch, dv, float(self.tail_bound), inverse=True, want_logdet=False)[0]
        return _spline_torch(self, input, inverse=True)[0]

    def logdet(self, input, context=None):
        return self.activation_and_logdet(input, context)[1]


def _spline_torch(layer, input, inverse):
    """The spline on torch expressions (reference semantics, activations.py:126-217 over
    splines/rational_quadratic.py:68-175): shared weights, or one set of knots per element (individual_weights: the
    MNIST Glow's activation -- per-element tables, gathered along the bin axis)."""
    tb = float(layer.tail_bound)
    inside = (input >= -tb) & (input <= tb)
    orig = input
    input = torch.where(inside, input, torch.zeros_like(input))  # (the tails are the identity: keep their lanes finite)
    if layer.individual_weights:
        cw, ch, dv = spline_tables(layer.unnormalized_widths, layer.unnormalized_heights, layer.unnormalized_derivatives, tb)
        cw, ch, dv = (t.expand(input.shape[0], *t.shape[1:]) for t in (cw, ch, dv))
        knots = ch if inverse else cw
        edges = torch.cat([knots[..., :-1], knots[..., -1:] + 1e-6], dim=-1)
        k = (torch.sum(input[..., None] >= edges, dim=-1) - 1).clamp(0, layer.n_bins - 1)[..., None]
        pick = lambda t, o: torch.gather(t, -1, k + o)[..., 0]  # noqa: E731
        a, b, c, e, d0, d1 = pick(cw, 0), pick(cw, 1), pick(ch, 0), pick(ch, 1), pick(dv, 0), pick(dv, 1)
    else:
        cw, ch, dv = layer._tables()
        knots = ch if inverse else cw
        edges = knots.clone()
        edges[-1] = edges[-1] + 1e-6
        k = (torch.sum(input[..., None] >= edges, dim=-1) - 1).clamp(0, layer.n_bins - 1)
        a, b, c, e, d0, d1 = cw[k], cw[k + 1], ch[k], ch[k + 1], dv[k], dv[k + 1]
    w, h = b - a, e - c
    delta = h / w
    if inverse:
        r = input - c
        s = d0 + d1 - 2 * delta
        qa = r * s + h * (delta - d0)
        qb = h * d0 - r * s
        qc = -delta * r
        root = (2 * qc) / (-qb - torch.sqrt(qb.pow(2) - 4 * qa * qc))
        out = root * w + a
        t1 = root * (1 - root)
        den = delta + s * t1
        dnum = delta.pow(2) * (d1 * root.pow(2) + 2 * delta * t1 + d0 * (1 - root).pow(2))
        lad = -(torch.log(dnum) - 2 * torch.log(den))
    else:
        theta = (input - a) / w
        t1 = theta * (1 - theta)
        num = h * (delta * theta.pow(2) + d0 * t1)
        den = delta + (d0 + d1 - 2 * delta) * t1
        out = c + num / den
        dnum = delta.pow(2) * (d1 * theta.pow(2) + 2 * delta * t1 + d0 * (1 - theta).pow(2))
        lad = torch.log(dnum) - 2 * torch.log(den)
    out = torch.where(inside, out, orig)
    lad = torch.where(inside, lad, torch.zeros_like(lad))
    return out, lad.flatten(start_dim=1).sum(dim=-1)


class Identity(FlowActivationLayer):
    def activation(self, input, context=None):
        return input

    def act_prime(self, input, context=None):
        return torch.ones_like(input)

    def reverse(self, input, context=None):
        return input
