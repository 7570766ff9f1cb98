"""The configs[4] model (if_multiGPU_imagenet32.py's configuration: L = 3, K = 48, width 256, as configured: no ActNorm;
--actnorm switches it on) at a rank's shard of the batch of 100 over eight ranks: ms per training step, eager
(--graph-only skips it) and as one captured graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.experiments.if_glow_imagenet32 import DEFAULT_CONFIG as cfg, create_model
from inf.train.step import TrainStep, bits_per_dim
dev = torch.device("cuda:0")
for graph in ((True,) if "--graph-only" in sys.argv else (False, True)):
    torch.manual_seed(5)
    model = create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                         if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                         coupling_width=cfg["coupling_width"], activation=cfg["activation"], actnorm="--actnorm" in sys.argv or cfg["actnorm"],
                         split_prior=cfg["split_prior"]).to(dev)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=graph,
                     batch_cond_prep=os.environ.get("BATCH_PREP", "1") == "1")
    x = torch.randint(0, 256, (13, 3, 32, 32), device=dev).float()
    for _ in range(5):
        loss = step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loss = step(x)
    torch.cuda.synchronize()
    print("graph=%s: %.1f ms per step, %.2f bits/dim, %d parameters" % (graph, (time.perf_counter() - t0) / 10 * 1e3,
          bits_per_dim(float(loss), 3 * 32 * 32), sum(p.numel() for p in model.parameters())))
