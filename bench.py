#!/usr/bin/env python3
"""Headline benchmark: inverse-conv forward + backward images/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One *step* = one pass of the hot path over one synthetic batch: `inverse` (x -> z = A^-1 x) plus the fused backward
(g, z -> dx, dW) at C=64, 32x32, K=3, fp32.  With N > 1 ranks (one process per GPU, RCCL) the batch is
sharded -- weak scaling: B=128 per rank (default); strong scaling: B=128 in total, 128/N per rank -- and the
step ends with the all-reduce of dW.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Run plainly with --gpus N > 1 (no WORLD_SIZE in the environment) this script starts its own N ranks through
torch.distributed.run, before anything touches a GPU, and relays rank 0's line (the reference: one process,
nn.DataParallel, inf/if_multiGPU_imagenet32.py:410-411).  --dry --backend gloo runs the same control flow on CPU tensors
without a kernel (the CPU test of the N > 1 path).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "inverse-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

B, C, HH, WW, K = 128, 64, 32, 32, 3
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* (f32 in/acc), dense
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (the kernels issue f16 MFMAs)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the configs[2] training-step figure")
    ap.add_argument("--flags", type=int, default=0, help="IFL_FLAG_* bits passed to the library")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with hipEvents")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--dry", action="store_true", help="no kernels, CPU tensors: exercises launch, sharding and the collective")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--workload", choices=("layer", "cifar_step", "imagenet32_step"), default="layer",
                    help="layer: the headline metric (BASELINE configs[1]); cifar_step / imagenet32_step: the training step of "
                         "configs[3] (batch 256) / configs[4] (batch 100), the batch sharded over the ranks")
    ap.add_argument("--no-graph", action="store_true", help="model workloads: eager steps instead of one captured graph per step")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a torch.distributed.run environment: start the N ranks (nothing here has touched a GPU) and
    relay rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout[-4000:])
        sys.exit(r.returncode or 1)
    print(lines[-1], flush=True)
    sys.exit(0)


def ref_init_weight(gen):
    """inf/layers/inv_conv.py:153-170: dirac + xavier_normal(gain=0.01), W[c,-1,-1,-1] = 1."""
    import torch
    w = torch.nn.init.dirac_(torch.empty(C, C, K, K))
    std = 0.01 * (2.0 / (2 * C * K * K)) ** 0.5
    w = w + torch.randn(C, C, K, K, generator=gen) * std
    w[:, -1, -1, -1] = 1.0
    return w


def algorithmic(tag, nb):
    """(flops, bytes) of ONE launch of the tagged kernel on nb images (per pixel-image:
    scan 2*(8C^2 + C(C-1)/2) flop and read+write of one activation; dW 2*9*C^2 flop, two reads)."""
    npix = nb * HH * WW
    act = npix * C * 4
    if tag == "scan":
        return 2.0 * (8 * C * C + C * (C - 1) / 2) * npix, 2.0 * act
    if tag == "wgrad":
        return 2.0 * K * K * C * C * npix, 2.0 * act
    if tag == "conv":
        return 2.0 * (8 * C * C + C * (C + 1) / 2) * npix, 2.0 * act
    return 0.0, 0.0


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(w, budget_s=10.0, budget_1t_s=6.0):
    """The CPU oracle (C restatement of the reference's exact solver, fp32, OpenMP over the batch as the reference's
    commented prange(batchsize), inverse_op_cython.pyx:35) timed on this host on a bounded sample of the same workload:
    inverse + dy + dw -- on all host cores, and on one thread (the reference ships single-threaded,
    inverse_op_cython.pyx:32).  A reported baseline, not the target."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    cores = host_cores()
    rng = np.random.default_rng(0)
    wn = w.numpy().astype(np.float32)

    def run(nthreads, nb, budget):
        total_img, total_t = 0, 0.0
        while total_t < budget:
            x = rng.standard_normal((nb, C, HH, WW)).astype(np.float32)
            g = rng.standard_normal((nb, C, HH, WW)).astype(np.float32)
            t0 = time.perf_counter()
            z = O.inverse(x, wn, nthreads=nthreads)
            u = O.dy(g, wn, nthreads=nthreads)
            O.dw(z, u, (K, K), nthreads=nthreads)
            total_t += time.perf_counter() - t0
            total_img += nb
        return total_img, total_t

    n_all, t_all = run(cores, max(cores, 8), budget_s)
    n_one, t_one = run(1, 1, budget_1t_s)
    return {"value": n_all / t_all, "unit": "images/s", "cores": cores, "kind": "port",
            "one_thread": {"value": n_one / t_one, "unit": "images/s", "cores": 1},
            "sample": "the C oracle (oracle/liboracle.so, fp32) on the same B=128,C=64,32x32,K=3 workload (inverse+dy+dw): "
                      "%d images in %.1f s with %d OpenMP threads, %d images in %.1f s with one thread"
                      % (n_all, t_all, cores, n_one, t_one)}


def accuracy(H, w, x, g, flags, nimg=3):
    """Relative L2 errors of z, dx, dW on a sub-batch of the bench's own inputs against the fp64 oracle, and the absolute
    error of log|det A| (exactly 0 for the unit diagonal).  Outside the timed region."""
    import numpy as np
    import torch
    from oracle import oracle as O
    O.build()
    xs, gs = x[:nimg].contiguous(), g[:nimg].contiguous()
    z = H.inverse(xs, w, "TL", flags)
    dx, dw, _ = H.backward(gs, z, w, "TL", flags)
    _, ld = H.forward(z, w, "TL", flags, want_logdet=True)
    x64, g64, w64 = xs.double().cpu().numpy(), gs.double().cpu().numpy(), w.double().cpu().numpy()
    nt = host_cores()
    z_o = O.inverse(x64, w64, nthreads=nt)
    u_o = O.dy(g64, w64, nthreads=nt)
    dw_o = O.dw(z_o, u_o, (K, K), nthreads=nt)

    def rel(a, b):
        return float(np.linalg.norm(a.double().cpu().numpy().ravel() - b.ravel()) / np.linalg.norm(b.ravel()))

    return {"rel_err_z": rel(z, z_o), "rel_err_dx": rel(dx, u_o), "rel_err_dw": rel(dw, dw_o),
            "logdet_abs_err": float(ld.abs().max().item()), "images": nimg,
            "reference": "oracle/ (fp64 restatement of inf/utils/solve_mc.py:88-114 and the adjoint / outer-product forms)"}


def train_step_figure(dev, steps=20, warmup=5):
    """BASELINE configs[2]: if_glow_mnist (L = 2, K = 16; inf/experiments/if_glow_mnist.py:33-190) full FlowSequential training
    step -- loss, backward, clip, Adam -- under bf16 autocast on synthetic uniform-dequantised 28x28x1 images, batch 100:
    ms per step and bits/dim (random-init weights: the bits/dim is that of an untrained model).  Not part of `value`."""
    import torch
    from inf.experiments.if_glow_mnist import DEFAULT_CONFIG as cfg, create_model
    from inf.train.step import TrainStep, bits_per_dim
    torch.manual_seed(3)
    model = create_model(num_blocks=cfg["num_blocks"], block_size=cfg["block_size"], coupling_width=cfg["coupling_width"],
                         n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"]).to(dev)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True)
    x = torch.randint(0, 256, (cfg["batch_size"], 1, 28, 28), device=dev).float()
    for _ in range(warmup):
        loss = step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"config": "configs[2]: if_glow_mnist L=2, K=16, batch 100, 28x28x1, bf16 autocast, Adam, synthetic uniform-dequantised data; "
                      "the step (forward, backward, clip, Adam) replayed as one captured graph after %d eager steps" % step.graph_warmup,
            "ms_per_step": ms, "images_per_s": cfg["batch_size"] / (ms * 1e-3), "bits_per_dim": bits_per_dim(float(loss), 28 * 28),
            "parameters": sum(p.numel() for p in model.parameters()), "steps": steps}


def cifar_step_figure(dev, steps=10, warmup=5):
    """BASELINE configs[3]: if_glow_cifar (inf/experiments/if_glow_cifar.py:28-190: L = 2, K = 16, 3x3 inverse-flow layers,
    shared splines, coupling width 128) training step at one rank's shard of the batch of 256 over eight GPUs (32 images),
    bf16 autocast, synthetic uniform-dequantised 32x32x3 images, as one captured graph.  As configured: no ActNorm (the
    layers start as the identity map, inf/layers/inv_conv.py _init_weight).  Not part of `value`."""
    import torch
    from inf.experiments.if_glow_cifar import DEFAULT_CONFIG as cfg, create_model
    from inf.train.step import TrainStep, bits_per_dim
    torch.manual_seed(4)
    model = create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                         if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                         coupling_width=cfg["coupling_width"], activation=cfg["activation"], actnorm=cfg["actnorm"],
                         split_prior=cfg["split_prior"]).to(dev)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True)
    nb = 32
    x = torch.randint(0, 256, (nb, 3, 32, 32), device=dev).float()
    for _ in range(warmup):
        loss = step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"config": "configs[3]: if_glow_cifar L=2, K=16, 3x3 inverse-flow layers, coupling width 128, 32 images per rank (256 over "
                      "8), 32x32x3, bf16 autocast, Adam, as configured (no ActNorm), synthetic data; one captured graph per step",
            "ms_per_step": ms, "images_per_s": nb / (ms * 1e-3), "bits_per_dim": bits_per_dim(float(loss), 3 * 32 * 32),
            "parameters": sum(p.numel() for p in model.parameters()), "steps": steps}


def imagenet32_step_figure(dev, steps=6, warmup=5):
    """BASELINE configs[4]: if_multiGPU_imagenet32 (inf/if_multiGPU_imagenet32.py:176-345: L = 3, K = 48, 3x3 inverse-flow
    layers, shared splines, coupling width 256, no ActNorm) training step at one rank's shard of its batch of 100 over eight
    GPUs (13 images), bf16 autocast, synthetic 32x32x3 images, one captured graph per step.  Not part of `value`."""
    import torch
    from inf.experiments.if_glow_imagenet32 import DEFAULT_CONFIG as cfg, create_model
    from inf.train.step import TrainStep, bits_per_dim
    torch.manual_seed(6)
    model = create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                         if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                         coupling_width=cfg["coupling_width"], activation=cfg["activation"], actnorm=cfg["actnorm"],
                         split_prior=cfg["split_prior"]).to(dev)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True)
    nb = 13
    x = torch.randint(0, 256, (nb, 3, 32, 32), device=dev).float()
    for _ in range(warmup):
        loss = step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"config": "configs[4]: if_multiGPU_imagenet32 L=3, K=48, 3x3 inverse-flow layers, coupling width 256, 13 images per rank "
                      "(100 over 8), 32x32x3, bf16 autocast, Adam, as configured (no ActNorm), synthetic data; one captured graph per step",
            "ms_per_step": ms, "images_per_s": nb / (ms * 1e-3), "bits_per_dim": bits_per_dim(float(loss), 3 * 32 * 32),
            "parameters": sum(p.numel() for p in model.parameters()), "steps": steps}


def wide_layer_figure(dev, steps=50, warmup=10):
    """BASELINE configs[4]'s channel count: one 3x3 inverse-conv layer of C = 256 on 8x8 at the per-GPU batch 16 (the 100
    images of inf/if_multiGPU_imagenet32.py:294 over 8 GPUs, rounded up), inverse + fused backward like the headline step.
    Device time by stream events.  Not part of `value`."""
    import torch
    import invflow_hip as H
    B_, C_, H_, W_, K_ = 16, 256, 8, 8, 3
    gen = torch.Generator().manual_seed(5)
    w = torch.zeros(C_, C_, K_, K_)
    w[:, :, -1, -1] = torch.eye(C_)
    w = (w + 0.01 * torch.randn(C_, C_, K_, K_, generator=gen)).to(dev)
    x = torch.randn(B_, C_, H_, W_, generator=gen).to(dev)
    g = torch.randn(B_, C_, H_, W_, generator=gen).to(dev)
    z, dx, dw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
    carry = H.new_carry(w)

    def step():
        H.inverse(x, w, "TL", 0, out=z, carry=carry)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)

    for _ in range(warmup):
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    return {"config": "one inverse-conv layer 3x3, C=256, 8x8, batch 16, fp32: inverse + fused backward", "ms_per_step": ms,
            "images_per_s": B_ / (ms * 1e-3), "steps": steps, "voided_launches": H.scan_voided(x.device)}


def committed_counters(kernel_us, world):
    """HBM-side bytes and matrix-pipe busy fraction of the scan from the rocprofv3 --pmc passes of this same command whose
    summaries profiles/LATEST.json names (PMC counters cannot be read from inside this process).  None when there is no
    pointer, or when the kernel it was taken on cannot be the one that just ran (launch time off by > 25 %: the boxes of
    this pool differ by +-10 % on the same binary -- 47 to 58 us for this kernel over round 3 -- and the bytes a launch moves
    do not depend on the clock)."""
    try:
        with open(os.path.join(ROOT, "profiles", "LATEST.json")) as f:
            latest = json.load(f)
        if world != 1 or abs(latest["scan_avg_us"] - kernel_us) > 0.25 * kernel_us:
            return None, None, None
        traffic = (latest["fetch_size_kib"] * latest["fetch_correction"] + latest["write_size_kib"]) * 1024.0
        return traffic, latest.get("mfma_busy_frac"), latest.get("source")
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, None


def model_workload(args, rank, local_rank, world):
    """BASELINE configs[3] / configs[4]: the training step of the 32x32x3 Glow with inverse-flow layers (if_glow_cifar: batch
    256; the multi-GPU ImageNet-32 model: batch 100), AS CONFIGURED, the batch sharded over the ranks (strong scaling: the
    reference hands the whole batch to nn.DataParallel, inf/if_multiGPU_imagenet32.py:410-411), bf16 autocast, one captured
    graph per step WITH the all-reduce of the flat gradient bucket inside it.  One JSON line: whole-job images/s, ms per step
    (max over ranks), the all-reduce alone.  --dry: the same control flow on CPU tensors (model construction, bucket,
    collective), no kernels."""
    import importlib
    import torch
    import torch.distributed as dist
    import data_parallel as dp
    from inf.train.step import TrainStep, bits_per_dim
    mod, total, tag = {"cifar_step": ("inf.experiments.if_glow_cifar", 256, "configs[3]: if_glow_cifar"),
                       "imagenet32_step": ("inf.experiments.if_glow_imagenet32", 100, "configs[4]: if_multiGPU_imagenet32")}[args.workload]
    m = importlib.import_module(mod)
    cfg = m.DEFAULT_CONFIG
    lo, hi = dp.shard_bounds(total, rank, world)
    nb = hi - lo
    torch.manual_seed(11)  # (the same model on every rank; broadcast below all the same)
    model = m.create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                           if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                           coupling_width=cfg["coupling_width"], n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"],
                           activation=cfg["activation"], actnorm=cfg["actnorm"], split_prior=cfg["split_prior"])
    nparam = sum(p.numel() for p in model.parameters())
    dev = torch.device("cpu") if args.dry else torch.device("cuda", local_rank)
    model = model.to(dev)
    dp.broadcast_parameters(model)
    torch.manual_seed(100 + rank)
    x = torch.randint(0, 256, (nb, 3, 32, 32), device=dev).float()
    clip = cfg["grad_clip_norm"]
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0 if clip is True else float(clip),
                     autocast=True, graph=not (args.no_graph or args.dry))
    loss = torch.zeros(())

    def one():
        if args.dry:  # no kernels: what a step does to the bucket between the backward and the optimizer
            step.bucket.flat.fill_(float(rank + 1))
            step.bucket.allreduce_mean()
            return torch.zeros(())
        return step(x)

    for _ in range(args.warmup):
        loss = one()
    if not args.dry:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.dry:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one()
    if not args.dry:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.dry:
        torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the collective alone (outside the timed region): the flat bucket, sum + scale
    ar_us = None
    if world > 1:
        n_ar = 20
        if args.dry:
            t1 = time.perf_counter()
            for _ in range(n_ar):
                step.bucket.allreduce_mean()
            ar_us = (time.perf_counter() - t1) / n_ar * 1e6
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            step.bucket.allreduce_mean()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n_ar):
                step.bucket.allreduce_mean()
            e1.record()
            torch.cuda.synchronize()
            ar_us = e0.elapsed_time(e1) / n_ar * 1e3
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        print(json.dumps({
            "metric": "%s training step images/sec (batch %d sharded over the ranks)" % (tag.split(": ")[1], total),
            "value": total * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 autocast (library layers: f16x3-split MFMA on f32 storage, f32 accumulate)",
            "data": "synthetic" if not args.dry else "none (dry run: no kernels)",
            "config": {"workload": "%s L=%d, K=%d, coupling width %d, as configured (actnorm=%s), batch %d = %d per rank, 32x32x3, Adam, "
                                   "%s" % (tag, cfg["num_blocks"], cfg["block_size"], cfg["coupling_width"], cfg["actnorm"], total, nb,
                                           ("one captured graph per step" if world == 1 else "two captured graphs per step around the bucket all-reduce") if step.graph else "eager steps"),
                       "per_rank_batch": nb, "parameters": nparam, "bucket_bytes": int(step.bucket.flat.numel()) * 4},
            "allreduce_us": ar_us,
            "bits_per_dim": None if args.dry else bits_per_dim(float(loss), 3 * 32 * 32),
            "scaling_curve": "this line is ONE N; the 1/2/4/8 curve is measured by the driver from such lines (SCALE_rNN.json), "
                             "or absent when no multi-GPU node was available",
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    import data_parallel as dp

    rank, local_rank, world = dp.init(args.backend or ("gloo" if args.dry else None))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, world))
    if args.workload != "layer":
        return model_workload(args, rank, local_rank, world)
    nb = B if args.scaling == "weak" else dp.shard_bounds(B, rank, world)[1] - dp.shard_bounds(B, rank, world)[0]
    nb_total = B * world if args.scaling == "weak" else B
    gen = torch.Generator().manual_seed(0)
    w_host = ref_init_weight(gen)
    torch.manual_seed(1 + rank)

    if args.dry:
        # control flow only: sharding, the collective, max-over-ranks timing, rank 0's line
        dw = torch.full((C, C, K, K), float(rank + 1))
        for _ in range(args.warmup):
            dp.allreduce_mean_(dw)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dp.allreduce_mean_(dw)
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "dry run (no kernels)", "value": nb_total * args.steps / float(t.item()), "unit": "images/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                              "per_rank_batch": nb, "dw_mean": float(dw.mean().item())}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    import invflow_hip as H
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    H.lib()
    x = torch.randn(nb, C, HH, WW, device=dev)
    g = torch.randn(nb, C, HH, WW, device=dev)
    w = w_host.to(dev)
    z = torch.empty_like(x)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    carry = H.new_carry(w)  # forward -> backward side channel of a step (what the autograd ctx carries)

    def step():
        H.inverse(x, w, "TL", args.flags, out=z, carry=carry)
        H.backward(g, z, w, "TL", args.flags, dx_out=dx, dw_out=dw, carry=carry)
        if world > 1:
            dp.allreduce_mean_(dw)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # ---- outside the timed region ---------------------------------------------------------------------------------------
    # per-step device times (events recorded on the launch stream between the steps: no host synchronisation inside)
    nper = max(100, min(args.steps, 400))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nper + 1)]
    evs[0].record()
    for i in range(nper):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nper))
    # per-kernel device times (roofline leg): hipEvents recorded by the library around each tagged launch, on the launch
    # stream; an event pair per launch costs ~6 us, so these steps are not part of `value`
    prof = {}
    if not args.no_kernel_events:
        H.profile_enable(True)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        H.profile_enable(False)
        prof = H.profile_collect()
    # "also report forward + log-det (z -> x^) separately"
    xh = torch.empty_like(z)
    for _ in range(3):
        H.forward(z, w, "TL", args.flags, out=xh, want_logdet=True)
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    for _ in range(20):
        H.forward(z, w, "TL", args.flags, out=xh, want_logdet=True)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - tf0) / 20 * 1e3

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = nb_total * args.steps / elapsed
        roofline = None
        if prof and any(v[1] for v in prof.values()):
            dom = max(prof, key=lambda k_: prof[k_][0])  # the tag with the largest device time
            ms, n = prof[dom]
            avg_s = ms / max(n, 1) * 1e-3
            flops, nbytes = algorithmic(dom, nb)
            achieved = flops / avg_s / 1e12
            traffic, mfma_busy, src = committed_counters(avg_s * 1e6, world) if dom == "scan" else (None, None, None)
            roofline = {
                # the path is a dense CxC contraction (166 flop/B): MFMA-bound.  achieved = ALGORITHMIC flops /
                # measured launch time; the kernels issue 3 f16 MFMAs per algorithmic product (split fp16, fp32 accumulate),
                # so the peak is the dense f16 MFMA peak.
                "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes": nbytes, "mfma_busy_frac": mfma_busy,
                "issued_tflops": 3.0 * achieved, "frac_issued": 3.0 * achieved / MFMA_F16_PEAK_TFLOPS,
                "frac_vs_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
                "avg_launch_us": avg_s * 1e6, "launches": n,
                "hbm_achieved_GBps": nbytes / avg_s / 1e9, "hbm_frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBPS,
                "step_hbm_frac": (5.0 * nb * C * HH * WW * 4) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "step_mfma_frac": (2 * algorithmic("scan", nb)[0] + algorithmic("wgrad", nb)[0]) / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "per_kernel_us": {k_: (v[0] / max(v[1], 1) * 1e3) for k_, v in prof.items() if v[1]},
            }
        acc = accuracy(H, w, x, g, args.flags)
        out = {
            "metric": "inverse-conv fwd+bwd images/sec @ B=128,C=64,32x32; log-det rel-err",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16x3-split (f32 in/out, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: single inverse-conv layer 3x3, C=64, 32x32, batch %d per GPU (%s scaling: %d in "
                                   "total), fp32: inverse (x->z) + fused backward (g,z->dx,dW)" % (nb, args.scaling, nb_total)
                                   + ("; dW all-reduce over RCCL" if world > 1 else ""),
                       "B": nb, "C": C, "H": HH, "W": WW, "K": K, "logdet_abs_err": acc["logdet_abs_err"]},
            "accuracy": acc,
            "step_ms_percentiles": {"p10": per[int(0.1 * (nper - 1))], "p50": per[(nper - 1) // 2], "p90": per[int(0.9 * (nper - 1))],
                                    "steps": nper, "what": "device time of single steps (stream events), outside the timed region"},
            "roofline": roofline,
            "forward_logdet": {"ms": fwd_ms, "images_per_s": nb / (fwd_ms * 1e-3),
                               "what": "ifl_forward_f32: z -> x^ = A z and log|det A| (the layer's reverse), per rank, not part of value"},
        }
        if world == 1 and not args.no_train_step:
            out["train_step"] = train_step_figure(dev)
            out["train_step_cifar"] = cifar_step_figure(dev)
            out["train_step_imagenet32"] = imagenet32_step_figure(dev)
            out["wide_layer"] = wide_layer_figure(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
