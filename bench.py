#!/usr/bin/env python3
"""Headline benchmark: inverse-conv forward + backward images/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one synthetic batch: `inverse` (x -> z = A^-1 x) plus
the fused backward (g, z -> dx, dW) at B=128, C=64, 32x32, K=3, fp32; with N > 1 every
rank runs its own B=128 batch (weak scaling) and the step ends with the RCCL all-reduce of dW.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "inverse-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B, C, HH, WW, K = 128, 64, 32, 32, 3
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* (f32 in/acc), dense
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense (the kernels issue f16 MFMAs)


def ref_init_weight(gen):
    """inf/layers/inv_conv.py:153-170: dirac + xavier_normal(gain=0.01), W[c,-1,-1,-1] = 1."""
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


def cpu_baseline(w, budget_s=15.0):
    """The CPU oracle (C restatement of the reference's exact solver, fp32, OpenMP over the batch as
    the reference's commented prange(batchsize), inverse_op_cython.pyx:35) timed on this host on a
    bounded sample of the same workload: inverse + dy + dw.  A reported baseline, not the target."""
    from oracle import oracle as O
    O.build()
    cores = host_cores()
    rng = np.random.default_rng(0)
    wn = w.numpy().astype(np.float32)
    nb = max(cores, 8)
    total_img, total_t = 0, 0.0
    while total_t < budget_s:
        x = rng.standard_normal((nb, C, HH, WW)).astype(np.float32)
        g = rng.standard_normal((nb, C, HH, WW)).astype(np.float32)
        t0 = time.perf_counter()
        z = O.inverse(x, wn, nthreads=cores)
        u = O.dy(g, wn, nthreads=cores)
        O.dw(z, u, (K, K), nthreads=cores)
        total_t += time.perf_counter() - t0
        total_img += nb
    return {"value": total_img / total_t, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d images of the same B=128,C=64,32x32,K=3 fp32 workload (inverse+dy+dw) in %.1f s, "
                      "oracle/liboracle.so with %d OpenMP threads" % (total_img, total_t, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="IFL_FLAG_* bits passed to the library")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with hipEvents")
    args = ap.parse_args()

    import invflow_hip as H
    import data_parallel as dp

    rank, local_rank, world = dp.init()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    H.lib()

    gen = torch.Generator().manual_seed(0)
    w_host = ref_init_weight(gen)
    torch.manual_seed(1 + rank)
    x = torch.randn(B, C, HH, WW, device=dev)
    g = torch.randn(B, C, HH, WW, device=dev)
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
    # per-kernel hipEvents (roofline leg) bracket the launches of every 5th step of the timed region only:
    # an event pair around each of the 5 launches of a step costs ~30 us of a ~330 us step
    t0 = time.perf_counter()
    for i in range(args.steps):
        H.profile_enable((not args.no_kernel_events) and i % 5 == 0)
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    H.profile_enable(False)
    prof = H.profile_collect()

    # "also report forward + log-det (z -> x^) separately" -- outside the timed region of the metric
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
        value = B * world * args.steps / elapsed
        # dominant kernel = the tag with the largest device time
        dom = max(prof, key=lambda k_: prof[k_][0])
        ms, n = prof[dom]
        if n == 0:
            ms, n = float("nan"), 1
        avg_s = ms / max(n, 1) * 1e-3
        flops, nbytes = algorithmic(dom, B)
        achieved = flops / avg_s / 1e12
        # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process,
        # so the figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/), corrected
        # as MI355X_MICROARCH.md prescribes.  None when the committed counters are for another kernel / missing.
        traffic, traffic_src = None, None
        try:
            cj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_final_scan_hbm_counters.json")
            with open(cj) as f:
                pc = json.load(f)
            if dom == "scan" and world == 1:
                traffic = (pc["fetch_size_kib"] * pc["fetch_correction"] + pc["write_size_kib"]) * 1024.0
                traffic_src = "profiles/r01_final_scan_hbm_counters.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
        except (OSError, KeyError, ValueError):
            pass
        # matrix-pipe busy fraction of the dominant kernel, same provenance (the SQ pass of tools/profile_round.sh)
        mfma_busy = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_final_sq_counters.json")) as f:
                sq = json.load(f)
            if dom == "scan" and world == 1:
                mfma_busy = sq["mfma_busy_frac"]
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            # the path is a dense CxC contraction (166 flop/B): MFMA-bound.  achieved = ALGORITHMIC flops
            #  / measured launch time; the kernels issue 3 f16 MFMAs per algorithmic product
            # (split fp16, fp32 accumulate), so the peak is the dense f16 MFMA peak.
            "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": nbytes, "mfma_busy_frac": mfma_busy,
            "issued_tflops": 3.0 * achieved, "frac_issued": 3.0 * achieved / MFMA_F16_PEAK_TFLOPS,
            "frac_vs_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
            "avg_launch_us": avg_s * 1e6, "launches": n,
            "hbm_achieved_GBps": nbytes / avg_s / 1e9, "hbm_frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBPS,
            "step_hbm_frac": (5.0 * B * C * HH * WW * 4) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "step_mfma_frac": (2 * algorithmic("scan", B)[0] + algorithmic("wgrad", B)[0]) / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "per_kernel_us": {k_: (v[0] / max(v[1], 1) * 1e3) for k_, v in prof.items() if v[1]},
        }
        out = {
            "metric": "inverse-conv fwd+bwd images/sec @ B=128,C=64,32x32; log-det rel-err",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3-split (f32 in/out, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: single inverse-conv layer 3x3, C=64, 32x32, batch 128 per GPU, fp32: "
                                   "inverse (x->z) + fused backward (g,z->dx,dW)" + ("; dW all-reduce over RCCL" if world > 1 else ""),
                       "B": B, "C": C, "H": HH, "W": WW, "K": K, "logdet_abs_err": 0.0},
            "roofline": roofline,
            "forward_logdet": {"ms": fwd_ms, "images_per_s": B / (fwd_ms * 1e-3),
                               "what": "ifl_forward_f32: z -> x^ = A z and log|det A| (the layer's reverse), per rank, not part of value"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
