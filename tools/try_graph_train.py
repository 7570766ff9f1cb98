"""configs[2] training step, eager and as one captured graph: ms per step and the loss sequence of both (same seed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.experiments.if_glow_mnist import DEFAULT_CONFIG as cfg, create_model
from inf.train.step import TrainStep
dev = torch.device("cuda:0")
for graph in ((True,) if "--graph-only" in sys.argv else (False, True)):
    torch.manual_seed(3)
    model = create_model(num_blocks=cfg["num_blocks"], block_size=cfg["block_size"], coupling_width=cfg["coupling_width"],
                         n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"]).to(dev)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=graph)
    x = torch.randint(0, 256, (cfg["batch_size"], 1, 28, 28), device=dev).float()
    losses = [float(step(x)) for _ in range(6)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        loss = step(x)
    torch.cuda.synchronize()
    print("graph=%s: %.2f ms per step; losses %s ... %.3f" % (graph, (time.perf_counter() - t0) / 20 * 1e3, ["%.2f" % v for v in losses], float(loss)))
