"""One training step of a FlowSequential model -- the part of the reference trainer the hot path sits in
(inf/train/experiment.py:160-195 get_loss, :272-311 the batch loop body, :15-17 clear_grad), without its harness (data
loaders, wandb, checkpoints, plots).

    loss = get_loss(model, x)                      -(log p + log-det), NaN -> 0, summed over the batch / len(x)
    step = TrainStep(model, optimizer, **config)   zero_grad -> loss -> backward -> [add_recon_grad] -> [clip] ->
    loss = step(x)                                 [all-reduce of the flat gradient bucket] -> optimizer.step()

Data parallel: one process per GPU; the parameters' gradients are views of ONE flat fp32 bucket
(data_parallel.GradBucket) that is all-reduced (mean) once per step over RCCL -- the reference wraps the model in
nn.DataParallel (inf/if_multiGPU_imagenet32.py:410-411).  With autocast=True the model's forward runs under bf16 autocast
(BASELINE configs[2]): the couplings' conditioner convolutions take bf16, the library layers cast to fp32 at their
boundary (inf/layers/*.py custom_fwd), the loss and the optimizer state stay fp32.
"""
import math

import torch

import data_parallel as dp
from inf.layers.inv_conv import inv_flow_with_pad


def get_loss(model, x):
    """experiment.py:160-195: mean negative log-likelihood of the batch in nats, NaN entries replaced by 0."""
    _, logp = model.forward(x)
    lossval = -logp
    lossval = torch.where(lossval != lossval, torch.zeros_like(lossval), lossval)
    return lossval.sum() / len(x)


def bits_per_dim(loss_nats, n_dims):
    """nats per image -> bits per dimension (the unit of BASELINE configs[2]'s "bits/dim")"""
    return float(loss_nats) / (n_dims * math.log(2.0))


def clear_grad(module):
    """experiment.py:15-17: the mask of the layers with padding orders, applied to their gradients"""
    if isinstance(module, inv_flow_with_pad):
        module.reset_gradients()


class TrainStep:
    def __init__(self, model, optimizer, add_recon_grad=False, grad_clip_norm=None, grad_clip=None, clear_grads=False,
                 autocast=False, bucket=True):
        self.model, self.optimizer = model, optimizer
        self.add_recon_grad, self.grad_clip_norm, self.grad_clip = add_recon_grad, grad_clip_norm, grad_clip
        self.clear_grads, self.autocast = clear_grads, autocast
        # every parameter's .grad is a view of one flat buffer: zeroing and the all-reduce are one operation each
        self.bucket = dp.GradBucket(model.parameters()) if bucket else None

    def __call__(self, x):
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast and x.is_cuda):
            lossval = get_loss(self.model, x)
        lossval.backward()
        if self.add_recon_grad:  # experiment.py:284-285 (the SelfNormConv layers' reconstruction term)
            self.model.add_recon_grad()
        if self.clear_grads:  # experiment.py:255 (the reference does this on its 'test' branch only)
            self.model.apply(clear_grad)
        if self.bucket is not None:
            self.bucket.allreduce_mean()
        if self.grad_clip_norm is not None:  # experiment.py:287-289
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm)
        if self.grad_clip:  # experiment.py:292-296: the reference clamps the PARAMETERS of layers that have a gradient
            for p in self.model.parameters():
                if p.grad is not None:
                    p.data.clamp_(-self.grad_clip, self.grad_clip)
        self.optimizer.step()
        return lossval.detach()
