"""The configs[3] training step (bench.py cifar_step_figure) alone, for a kernel trace: python tools/prof_cifar_step.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import bench
print(bench.cifar_step_figure(torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 40))
