import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cnc_amd.mlp import FusedMLPForward, Linear
dev = torch.device("cuda:0")
dims = (255, 160, 80)
seq = nn.Sequential(Linear(255, 160), nn.ReLU(inplace=True), Linear(160, 80)).to(dev)
fused = FusedMLPForward(seq)
x = torch.randn(1 << 20, 256, device=dev)[:, :255]      # row stride 256, as the field feeds it
with torch.no_grad():
    for _ in range(3):
        seq(x); fused(x)
torch.cuda.synchronize()
