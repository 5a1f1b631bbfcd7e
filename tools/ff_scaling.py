#!/usr/bin/env python3
"""kernel time of the GNT kernels against the number of rows (fixed cost vs per-tile cost)"""
import pathlib, sys
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from pgdvs_amd.models.gnt.models.transformer_network import GNT
dev = "cuda:0"
net = GNT(netwidth=64, transformer_depth=1).to(dev).eval()
layer = net.view_selftrans[0]
for R in (256, 512, 1024, 2048, 4096):
    q = torch.randn(R, 256, 64, device=dev)
    with torch.no_grad():
        net._ray_layer(layer, q, True); torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3): net._ray_layer(layer, q, True)
            torch.cuda.synchronize()
    row = {e.key.split("(")[0].split("::")[-1]: e.device_time_total / e.count for e in prof.key_averages() if "gnt_" in e.key}
    print(f"rays={R:5d} rows={R*256:8d}: " + "  ".join(f"{k}={v:8.1f}us" for k, v in sorted(row.items())))
