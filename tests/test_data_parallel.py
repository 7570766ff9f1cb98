"""CPU: the batch-sharded data-parallel path (one process per rank, gloo, world size 2): shard
bounds, flat gradient bucket, all-reduce mean -- the N>1 logic bench.py and a training loop use."""
import os
import subprocess
import sys

import torch

from conftest import PKG, ROOT

import data_parallel as dp


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 128, 130):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    x = torch.arange(10).view(10, 1)
    assert torch.equal(dp.shard_batch(x, 1, 3), x[4:7])


def test_grad_bucket_views():
    lin = torch.nn.Linear(3, 2)
    b = dp.GradBucket(lin.parameters())
    assert b.flat.numel() == 8 and lin.weight.grad.data_ptr() == b.flat.data_ptr()
    lin(torch.ones(1, 3)).sum().backward()
    assert float(b.flat.abs().sum()) > 0
    b.zero()
    assert float(lin.weight.grad.abs().sum()) == 0.0
    assert b.allreduce_mean() is None  # single process: no-op


WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, torch.distributed as dist
import data_parallel as dp
rank, local, world = dp.init("gloo")
assert world == 2 and dist.get_world_size() == 2
torch.manual_seed(0)
model = torch.nn.Conv2d(4, 4, 3, padding=1, bias=False)
dp.broadcast_parameters(model)
bucket = dp.GradBucket(model.parameters())
x = torch.randn(8, 4, 6, 6)             # same full batch on both ranks (same seed)
ref = torch.nn.Conv2d(4, 4, 3, padding=1, bias=False)
ref.load_state_dict(model.state_dict())
ref(x).pow(2).mean().backward()          # single-process gradient of the mean loss
xs = dp.shard_batch(x)                   # this rank's shard
assert xs.shape[0] == 4
bucket.zero()
model(xs).pow(2).mean().backward()       # per-rank mean
w = bucket.allreduce_mean(async_op=True)
w.wait()
assert torch.allclose(model.weight.grad, ref.weight.grad, atol=1e-6), (model.weight.grad - ref.weight.grad).abs().max()
t = torch.full((3,), float(rank + 1))
dp.allreduce_mean_(t)
assert torch.allclose(t, torch.full((3,), 1.5))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (PKG, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") == 2


INIT_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, torch.distributed as dist
import data_parallel as dp
from inf.layers.actnorm import ActNorm
from inf.layers.flowsequential import FlowSequential
from inf.train.losses import NegativeGaussianLoss
from inf.train.step import TrainStep
rank, local, world = dp.init("gloo")
torch.manual_seed(7)                       # same model on both ranks ...
model = FlowSequential(NegativeGaussianLoss((3, 4, 4)), ActNorm(3), ActNorm(3))
step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=1e-2))
torch.manual_seed(100 + rank)              # ... different shards: the data-dependent initialisation would differ
x = 3.0 * torch.randn(6, 3, 4, 4) + rank
loss = step(x)
for _ in range(2):
    step(3.0 * torch.randn(6, 3, 4, 4) + rank)
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [b.detach().float().reshape(-1) for b in model.buffers()])
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), (both[0] - both[1]).abs().max()   # initialised alike, stepped alike
assert all(int(m.initialized) == 1 for m in model.modules() if isinstance(m, ActNorm))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_first_batch_initialisation_is_rank_zeros_on_every_rank(tmp_path):
    """ActNorm's data-dependent initialisation is per shard; TrainStep makes rank 0's the common one before the
    first step, and the replicas stay identical under the averaged gradients."""
    script = tmp_path / "init_worker.py"
    script.write_text(INIT_WORKER % (PKG, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") == 2


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` run plainly starts two ranks by itself (before anything touches a GPU) and rank 0's line
    says n_gpus 2; strong scaling shards the batch of 128; a process group of the wrong size is refused.  --dry: the same
    control flow on CPU tensors over gloo, no kernel."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OMP_NUM_THREADS="1")
    bench = os.path.join(ROOT, "bench.py")
    for scaling, per_rank in (("weak", 128), ("strong", 64)):
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry", "--backend", "gloo",
                            "--scaling", scaling, "--master-port", "29537"], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["per_rank_batch"] == per_rank
        assert abs(line["dw_mean"] - 1.5) < 1e-6  # mean of the ranks' values 1 and 2: the collective ran
    # one process that claims to be a 1-rank group must not pass for --gpus 2
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry", "--backend", "gloo"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120)
    assert r.returncode != 0 and "process group has 1 ranks" in (r.stdout + r.stderr)


def test_bench_model_workloads_shard_the_configured_batch(tmp_path):
    """`bench.py --workload cifar_step|imagenet32_step --gpus 2 --backend gloo --dry`: BASELINE configs[3] / configs[4] -- the
    models as configured, the batch of 256 / 100 sharded over the ranks, the flat gradient bucket all-reduced -- control flow
    on CPU tensors (the models are built, nothing is launched)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(OMP_NUM_THREADS="1")
    bench = os.path.join(ROOT, "bench.py")
    for wl, total, per_rank, nparam_min in (("cifar_step", 256, 128, 600_000), ("imagenet32_step", 100, 50, 8_000_000)):
        r = subprocess.run([sys.executable, bench, "--workload", wl, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry", "--backend",
                            "gloo", "--master-port", "29541"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["per_rank_batch"] == per_rank
        assert line["config"]["parameters"] > nparam_min and line["config"]["bucket_bytes"] == 4 * line["config"]["parameters"]
        assert "actnorm=False" in line["config"]["workload"] and line["allreduce_us"] > 0 and "driver" in line["scaling_curve"]
