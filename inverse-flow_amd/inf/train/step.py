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
import time

import torch

import data_parallel as dp
from inf.layers.inv_conv import inv_flow_with_pad
from inf.layers.coupling import ConditionerPrep


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
    """graph=True: after `graph_warmup` eager steps the whole step -- forward, backward, clip, optimizer -- is captured into
    one HIP graph and every later call is a copy of the batch into the captured input plus one replay (the step of a
    small-image Glow is hundreds of short launches: host-bound when issued one by one).  Needs a fixed batch shape, an
    optimizer that can be captured (Adam/AdamW get capturable=True set here, before their first step) and the gradient
    bucket; every launch of the library is stream-ordered and allocation-free, so it captures as it is."""

    def __init__(self, model, optimizer, add_recon_grad=False, grad_clip_norm=None, grad_clip=None, clear_grads=False,
                 autocast=False, bucket=True, graph=False, graph_warmup=3, force_collective=False, conv_search=False,
                 fused_optimizer=True, gather_grads=None, flat_optimizer=None, batch_cond_prep=True):
        self.model, self.optimizer = model, optimizer
        # the weight images of all fused couplings in one launch at the head of the step (inf/layers/coupling.py)
        self.cond_prep = ConditionerPrep(model) if batch_cond_prep else None
        self.add_recon_grad, self.grad_clip_norm, self.grad_clip = add_recon_grad, grad_clip_norm, grad_clip
        self.clear_grads, self.autocast = clear_grads, autocast
        self.force_collective = force_collective  # issue the bucket's all-reduce in a one-rank process group too (tests)
        # every parameter's .grad is a view of one flat buffer: zeroing and the all-reduce are one operation each
        self.bucket = dp.GradBucket(model.parameters()) if bucket else None
        self._n_params = sum(1 for _ in model.parameters())
        self.graph, self.graph_warmup = graph, graph_warmup
        self.gather_grads = graph if gather_grads is None else gather_grads
        self._calls, self._captured, self._static_x, self._static_loss, self._stream = 0, None, None, None, None
        self._captured2, self._split = None, False
        self._ranks_agree = False  # data-dependent initialisations (ActNorm) made identical on every rank
        # conv_search=True: MIOpen times its solvers for the conditioners' convolutions in the warm-up steps instead of taking
        # its heuristic's pick (a process-wide torch switch).  Off by default: worth 0-5 % of the configs[2]/[3] steps, and
        # its picks -- and with them the last digits of the results -- vary from run to run (tools/time_trainsteps.py).
        if conv_search:
            torch.backends.cudnn.benchmark = True
        if graph:
            if self.bucket is None:
                raise ValueError("TrainStep(graph=True) needs the gradient bucket (the gradients' addresses must not move)")
            for group in optimizer.param_groups:
                if "capturable" in group:
                    group["capturable"] = True
            self._make_capturable()
            if fused_optimizer:
                self._fuse_optimizer()
        self._flat = None
        if graph if flat_optimizer is None else flat_optimizer:
            self._flatten_optimizer()

    def _make_capturable(self):
        """A captured optimizer step bakes every Python number it sees into its kernels.  The learning rate therefore
        becomes a device tensor (torch's capturable Adam/AdamW read it on the device): `set_lr` and schedulers that assign
        `group['lr']` through it change it in place and the next replay uses the new value (the reference's trainer changes
        it per batch during warm-up, experiment.py:197+).  Optimizer state restored from a checkpoint lives on the CPU
        (`step` counters): moved to the parameters' device here, which capturable optimizers require."""
        dev = None
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                dev = p.device
                break
            if dev is not None:
                break
        if dev is None or dev.type != "cuda":
            return
        for group in self.optimizer.param_groups:
            if "capturable" in group and not torch.is_tensor(group["lr"]):
                group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=dev)
        for st in self.optimizer.state.values():
            for k, v in list(st.items()):
                if torch.is_tensor(v) and v.device != dev:
                    st[k] = v.to(dev)

    def _fuse_optimizer(self):
        """Adam/AdamW over a model of a few hundred small tensors: torch's capturable foreach implementation divides every
        tensor by two device scalars (`_foreach_div_` with 0-dim divisors takes its slow path: two broadcast kernels per
        PARAMETER, 580 launches a step for the configs[3] model, a third of the step's dispatches -- tools/op_breakdown.py).
        The fused multi-tensor implementation is one launch per ~hundred tensors, takes the learning rate as a device tensor
        and can be captured: switched on here, before the optimizer's first step creates its state."""
        if not isinstance(self.optimizer, (torch.optim.Adam, torch.optim.AdamW)) or self.optimizer.state:
            return
        for group in self.optimizer.param_groups:
            if group.get("differentiable") or group.get("fused"):
                continue
            if all(p.is_cuda and torch.is_floating_point(p) for p in group["params"]):
                group["foreach"], group["fused"] = False, True

    def _flatten_optimizer(self):
        """Adam / AdamW as ONE elementwise pass (ifl_adam_flat_f32) over a flat parameter buffer laid out like the gradient
        bucket.  torch's fused multi-tensor Adam takes 36 tensors a launch: the configs[4] model's 1 300 tensors are 45
        launches of ~48 us (2.2 ms a step) for what is 235 MB of traffic.  The parameters are re-homed as views of the flat
        buffer (same values, same modules, same state_dict); exp_avg / exp_avg_sq are views of two more, entered in the
        optimizer's state so that its state_dict() stays a torch.optim one.  Parameters that share their storage with
        another parameter (Conv2dZero's bias and logs, as in the reference) keep it and a small fused optimizer of their
        own; their slots in the flat buffers are unused.  Anything else (several parameter groups, amsgrad, CPU, an
        optimizer with state, parameters outside the bucket) keeps the optimizer as it is."""
        import invflow_hip as H
        opt, bucket = self.optimizer, self.bucket
        if bucket is None or not isinstance(opt, (torch.optim.Adam, torch.optim.AdamW)) or opt.state or len(opt.param_groups) != 1:
            return
        g = opt.param_groups[0]
        params = [p for p in g["params"] if p.requires_grad]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or len(params) != len(bucket.params) or \
                any(a is not b for a, b in zip(params, bucket.params)) or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return
        dev = bucket.flat.device
        lr = g["lr"] if torch.is_tensor(g["lr"]) else torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
        g["lr"] = lr
        users = {}
        for p in params:
            users[p.untyped_storage().data_ptr()] = users.get(p.untyped_storage().data_ptr(), 0) + 1
        flat_p, m, v = torch.zeros_like(bucket.flat), torch.zeros_like(bucket.flat), torch.zeros_like(bucket.flat)
        step_t = torch.zeros((), dtype=torch.float32, device=dev)
        off, shared = 0, []
        for p in params:
            n = p.numel()
            if users[p.untyped_storage().data_ptr()] == 1 and p.is_contiguous():
                dst = flat_p[off:off + n].view_as(p)
                dst.copy_(p.data)
                p.data = dst
                opt.state[p] = {"step": step_t, "exp_avg": m[off:off + n].view_as(p), "exp_avg_sq": v[off:off + n].view_as(p)}
            else:
                shared.append(p)
            off += n
        rest = None
        if shared:
            rest = type(opt)(shared, lr=lr, betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], capturable=True, fused=True)
        self._flat = dict(p=flat_p, m=m, v=v, step=step_t, lr=lr, rest=rest, H=H, decoupled=isinstance(opt, torch.optim.AdamW))

    def _optimizer_step(self):
        f = self._flat
        if f is None:
            self.optimizer.step()
            return
        g = self.optimizer.param_groups[0]
        f["step"].add_(1.0)
        f["H"].adam_flat(f["p"], self.bucket.flat, f["m"], f["v"], f["lr"].reshape(1), f["step"].reshape(1), g["betas"][0], g["betas"][1],
                         g["eps"], g["weight_decay"], f["decoupled"])
        if f["rest"] is not None:
            f["rest"].step()

    def set_lr(self, lr):
        """change the learning rate of every parameter group -- in place when it is a device tensor (graph=True), so that a
        captured step picks it up"""
        for group in self.optimizer.param_groups:
            if torch.is_tensor(group["lr"]):
                group["lr"].fill_(float(lr))
            else:
                group["lr"] = float(lr)

    def _agree_on_init(self, x):
        """ActNorm's data-dependent initialisation (actnorm.py:21-27) sees a different shard on every rank.
        Before the first step every rank runs one forward pass without gradients -- the layers initialise themselves -- and
        then takes rank 0's parameters and buffers, so that the replicas start, and with the averaged gradients stay,
        identical.  (The reference's DataParallel re-broadcasts module 0's parameters on every step.)"""
        self._ranks_agree = True
        if dp.dist.is_available() and dp.dist.is_initialized() and dp.dist.get_world_size() > 1:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast and x.is_cuda):
                get_loss(self.model, x)
            dp.broadcast_parameters(self.model)

    def _collective_due(self):
        """a process group of more than one rank (or a test that wants the collective issued in a group of one)"""
        return self.bucket is not None and dp.dist.is_available() and dp.dist.is_initialized() and (
            dp.dist.get_world_size() > 1 or self.force_collective)

    def _sync_lr(self):
        """a scheduler that ASSIGNS group['lr'] (a Python number) instead of filling the device tensor: take the value into the
        tensor the captured / flat optimizer step reads and put the tensor back"""
        groups = self.optimizer.param_groups
        if getattr(self, "_lr_t", None) is None or len(self._lr_t) != len(groups):
            self._lr_t = [g["lr"] if torch.is_tensor(g["lr"]) and g["lr"].is_cuda else None for g in groups]
        for g, t in zip(groups, self._lr_t):
            if t is not None and g["lr"] is not t:
                t.fill_(float(g["lr"]))
                g["lr"] = t

    def __call__(self, x):
        if self.graph:
            self._sync_lr()
        if not self._ranks_agree:
            self._agree_on_init(x)
        if not self.graph or not x.is_cuda:
            return self._eager(x)
        if self._captured is None:
            # eager steps first (data-dependent inits, optimizer state, workspaces) -- on the stream the capture will use, so
            # that autograd's accumulation nodes and the library's per-stream scan state belong to it
            if self._stream is None:
                self._stream = torch.cuda.Stream(x.device)
            cur = torch.cuda.current_stream(x.device)
            if self._calls < self.graph_warmup:
                self._calls += 1
                self._stream.wait_stream(cur)
                with torch.cuda.stream(self._stream):
                    loss = self._eager(x)
                cur.wait_stream(self._stream)
                return loss
            self._static_x = x.clone()
            torch.cuda.synchronize()
            if self._collective_due():
                # ProcessGroupNCCL's watchdog retires a finished collective on its next poll (every 100 ms) by querying the
                # work's event -- and HIP refuses the query of an event whose STREAM is capturing by then
                # (hipErrorCapturedEvent ends the process; the warm-up's all-reduces were issued on the stream captured
                # below).  Seen once in a few dozen runs.  All work is finished here: leave the watchdog a few polls.
                time.sleep(0.5)
            # With a collective in the step the capture is cut in two AROUND it: [loss, backward] and [clip, optimizer], the
            # all-reduce of the bucket issued eagerly between the two replays.  (A captured RCCL all-reduce replays fine, but
            # ProcessGroupNCCL's watchdog queries the work's event, which was recorded in a capturing stream:
            # hipErrorCapturedEvent ends the process -- torch 2.10 + ROCm 7.0,
            # tests/test_hip_train.py::test_captured_step_with_the_bucket_all_reduce_inside.)
            self._split = self._collective_due()
            self._captured = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._captured, stream=self._stream):
                self._static_loss = self._backward_part(self._static_x)
                if not self._split:
                    self._update_part()
            if self._split:
                self._captured2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._captured2, stream=self._stream, pool=self._captured.pool()):
                    self._update_part()
            return self._replay()  # (capturing records the step without running it: this batch's step runs now)
        if x.shape != self._static_x.shape:
            raise ValueError("TrainStep(graph=True) was captured for batches of shape %s" % (tuple(self._static_x.shape),))
        self._static_x.copy_(x)
        return self._replay()

    def _replay(self):
        self._captured.replay()
        if self._split:
            self.bucket.allreduce_mean(force=self.force_collective)
            self._captured2.replay()
        return self._static_loss.clone()

    def _backward_part(self, x):
        gather = self.bucket is not None and self.gather_grads and x.is_cuda
        if gather:
            # backward into fresh tensors (autograd keeps the tensor a parameter's first gradient arrives in: no kernel),
            # then ONE multi-tensor copy into the flat bucket -- instead of one accumulation kernel per parameter into its
            # zeroed bucket view (290 launches a step for the configs[3] model)
            self.bucket.release()
        elif self.bucket is not None:
            self.bucket.zero()
        else:
            self.optimizer.zero_grad()
        if x.is_cuda and self.cond_prep is not None:
            self.cond_prep.fill()  # (the couplings' weight images: one launch for all of them)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast and x.is_cuda):
                lossval = get_loss(self.model, x)
            lossval.backward()
        finally:
            if self.cond_prep is not None:
                self.cond_prep.release()  # (the optimizer changes the kernels: the images are this step's only)
        if gather:
            self.bucket.gather()
        if self.add_recon_grad:  # experiment.py:284-285 (the SelfNormConv layers' reconstruction term)
            self.model.add_recon_grad()
        if self.clear_grads:  # experiment.py:255 (the reference does this on its 'test' branch only)
            self.model.apply(clear_grad)
        return lossval.detach()

    def _update_part(self):
        if self.grad_clip_norm is not None:  # experiment.py:287-289
            if self.bucket is not None and len(self.bucket.params) == self._n_params:
                # every gradient is a view of the flat bucket: its norm and the scaling are one kernel each, with the
                # arithmetic of clip_grad_norm_ (coef = max_norm / (norm + 1e-6), clamped to 1)
                norm = torch.linalg.vector_norm(self.bucket.flat)
                self.bucket.flat.mul_(torch.clamp(self.grad_clip_norm / (norm + 1e-6), max=1.0))
            else:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm)
        if self.grad_clip:  # experiment.py:292-296: the reference clamps the PARAMETERS of layers that have a gradient
            for p in self.model.parameters():
                if p.grad is not None:
                    p.data.clamp_(-self.grad_clip, self.grad_clip)
        self._optimizer_step()

    def _eager(self, x):
        lossval = self._backward_part(x)
        if self.bucket is not None:
            self.bucket.allreduce_mean(force=self.force_collective)
        self._update_part()
        return lossval
