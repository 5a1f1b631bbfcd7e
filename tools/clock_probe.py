#!/usr/bin/env python3
"""Sample the GPU engine clock / power (rocm-smi) while the GNT forward pass runs in a loop:
are the MFMA kernels clock-limited by the power cap?  (diagnostic, GPU box only)"""
import pathlib, subprocess, sys, threading, time
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import torch
from pgdvs_amd.models.gnt.models.transformer_network import GNT

dev = "cuda:0"
net = GNT(netwidth=64, transformer_depth=8).to(dev).eval()
R, S, V = 1024, 256, 24
rgb_feat = torch.randn(R, S, V, 35, device=dev); rd = torch.randn(R, S, V, 4, device=dev)
mask = (torch.rand(R, S, V, 1, device=dev) < 0.8).float(); pts = torch.randn(R, S, 3, device=dev); ray_d = torch.randn(R, 3, device=dev)
stop = False
def work():
    with torch.no_grad():
        while not stop:
            net(rgb_feat, rd, mask, pts, ray_d, ret_view_entropy=True, ret_view_std=True)
            torch.cuda.synchronize()
def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    return [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "junction"))]
print("idle:", *smi(), sep="\n  ")
t = threading.Thread(target=work); t.start()
time.sleep(1.5)
for i in range(4):
    print(f"under load #{i}:", *smi(), sep="\n  ")
    time.sleep(0.7)
stop = True; t.join()
