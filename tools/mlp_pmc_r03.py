"""MFMA-shaped work of the path under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace:
one training step of the radiance field's two networks (255->160->80 and 95->160->160->3, fp32, N = 2^18 rows: forward,
input gradients, split-K weight gradients — cnc_amd/mlp.py on hipBLASLt) and the hand-written fused evaluation
kernels (cnc_mlp_forward32 on both networks, N = 2^20).  tools/summarise_mfma_r03.py makes the table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from cnc_amd.mlp import FusedMLPForward, Linear, run_layers

dev = torch.device("cuda:0")
torch.manual_seed(0)
base = nn.Sequential(Linear(255, 160), nn.ReLU(inplace=True), Linear(160, 80)).to(dev)
head = nn.Sequential(Linear(95, 160), nn.ReLU(inplace=True), Linear(160, 160), nn.ReLU(inplace=True), Linear(160, 3)).to(dev)
N = 1 << 18
x = torch.randn(N, 256, device=dev)[:, :255].requires_grad_(True)
d = torch.randn(N, 16, device=dev)
for it in range(3):
    h = run_layers(base, x)
    rgb = run_layers(head, torch.cat([d, h[:, 1:]], -1))
    (rgb.sum() + h[:, 0].sum()).backward()
    for p in list(base.parameters()) + list(head.parameters()):
        p.grad = None
torch.cuda.synchronize()
fb, fh = FusedMLPForward(base, rows_per_wave=32), FusedMLPForward(head, rows_per_wave=32)
xe = torch.randn(1 << 20, 256, device=dev)[:, :255]
he = torch.randn(1 << 20, 96, device=dev)[:, :95]
with torch.no_grad():
    for _ in range(3):
        fb(xe); fh(he)
        run_layers(base, xe); run_layers(head, he)
torch.cuda.synchronize()
