"""ActNorm on the HIP library (reference: inf/layers/actnorm.py:5-92).

Same parameters (`translation`, `log_scale`, buffer `initialized`), same data-dependent initialisation on
the first forward, same outputs: forward -> ((x - t) exp(-log_scale), -H W sum log_scale), reverse ->
x exp(log_scale) + t.  4-D CUDA fp32 inputs run in libinvflow_hip (ifl_actnorm_f32 and its backward: one
pass over the activation each, the per-channel gradient sums in the same pass); anything else (2-D inputs of
ActNormFC, CPU tensors) takes the reference's torch expressions.
"""
import torch

import invflow_hip as H

from .activations import FlowActivationLayer

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


class _ActNormFn(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, translation, log_scale):
        x = x.contiguous()
        y, ld = H.actnorm(x, translation.contiguous(), log_scale.contiguous())
        ctx.save_for_backward(x, translation, log_scale)
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, translation, log_scale = ctx.saved_tensors
        gx, gt, gls = H.actnorm_backward(gy.contiguous(), None if gld is None else gld.contiguous(), x,
                                         translation.contiguous(), log_scale.contiguous())
        return gx, gt, gls


def _hip_ok(t):
    return t.dim() == 4 and t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16)


class ActNorm(FlowActivationLayer):
    def __init__(self, n_dims):
        super().__init__()
        self.n_dims = n_dims
        self.translation = torch.nn.Parameter(torch.zeros(n_dims))
        torch.nn.init.normal_(self.translation)
        self.log_scale = torch.nn.Parameter(torch.zeros(n_dims))
        torch.nn.init.normal_(self.log_scale)
        self.register_buffer('initialized', torch.tensor(0))
        # host copy of the flag: reading the buffer costs a device sync per call (and cannot be captured into a graph);
        # None = unknown (fresh module, or a state dict has just been loaded) -> read the buffer once
        self._init_seen = None

    def _is_initialized(self):
        if not self._init_seen:
            self._init_seen = bool(self.initialized)
        return self._init_seen

    def _load_from_state_dict(self, *args, **kwargs):
        self._init_seen = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _initialize(self, input):
        with torch.no_grad():
            if _hip_ok(input) and input.dtype == torch.float32:
                mean, log_std = H.actnorm_stats(input.contiguous())
            else:  # actnorm.py:21-26
                reduce_dims = [i for i in range(input.dim()) if i != 1]
                mean = torch.mean(input, dim=reduce_dims)
                log_std = torch.log(torch.std(input, dim=reduce_dims) + 1e-8)
            self.translation.data.copy_(mean)
            self.log_scale.data.copy_(log_std)
            self.initialized.fill_(1)
            self._init_seen = True

    def _views(self, input):
        shape = (1, -1, 1, 1) if input.dim() == 4 else (1, -1)
        return self.translation.view(shape), self.log_scale.view(shape)

    def forward(self, input, context=None):
        if not self._is_initialized():
            self._initialize(input)
        if _hip_ok(input):
            return _ActNormFn.apply(input, self.translation, self.log_scale)
        translation, log_scale = self._views(input)
        return (input - translation) * torch.exp(-log_scale), self.logdet(input, context)

    def reverse(self, input, context=None):
        assert self._is_initialized()
        if _hip_ok(input) and input.dtype in (torch.float32, torch.bfloat16) and not torch.is_grad_enabled():
            return H.actnorm(input.contiguous(), self.translation.contiguous(), self.log_scale.contiguous(), reverse=True)
        translation, log_scale = self._views(input)
        return input * torch.exp(log_scale) + translation

    def act_prime(self, input, context=None):
        return torch.exp(-self.log_scale)

    def logdet(self, input, context=None):
        B = input.size(0)
        ldj = -self.log_scale.sum().expand(B)
        if input.dim() == 4:
            ldj = ldj * input.size(2) * input.size(3)
        return ldj


class ActNormPlainLayer(ActNorm):
    def forward(self, *args, **kwargs):
        out, ldj = super().forward(*args, **kwargs)
        return out


class ActNormFC(ActNorm):
    """2-D inputs (B, D): the torch expressions above (actnorm.py:77-92)."""
