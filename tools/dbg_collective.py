"""the worker of tests/test_hip_train.py::test_captured_step_with_the_bucket_all_reduce_inside with TrainStep's
batch_cond_prep switch on the command line (0 / 1): a captured step in a one-rank RCCL group, development aid"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29548", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from test_hip_train import build
from inf.train.step import TrainStep
d = np.load(os.path.join(ROOT, "tests", "golden", "trainstep_glow_b6_8x8_L2K2.npz")); fixture = {k: d[k] for k in d.files}
x = torch.from_numpy(fixture["x"]).float().cuda()
batched = sys.argv[1] == "1"
seqs = []
for graph, force in ((True, False), (True, True)):
    torch.manual_seed(0)
    model = build(fixture)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), grad_clip_norm=1.0, graph=graph, graph_warmup=2,
                     force_collective=force, batch_cond_prep=batched)
    seqs.append([float(step(x)) for _ in range(8)])
    print("run", graph, force, "done", flush=True)
assert seqs[0] == seqs[1], seqs
dist.destroy_process_group()
print("batched=%s: ok" % batched)
