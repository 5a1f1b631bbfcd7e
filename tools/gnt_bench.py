#!/usr/bin/env python3
"""GNT sub-benchmark: the view-transformer contraction on the fp32 matrix cores.
Times one layer on a chunk of R rays x S samples x V views and reports TFLOP/s against the
fp32-MFMA peak (157.3 TFLOP/s); also the full GNT forward of a chunk (all layers)."""
import argparse, pathlib, sys, time
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import torch
from pgdvs_amd import ops
from pgdvs_amd.models.gnt.models.transformer_network import GNT

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=2048)
ap.add_argument("--samples", type=int, default=256)
ap.add_argument("--views", type=int, default=10)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--stats", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
net = GNT(netwidth=64, transformer_depth=a.depth).to(dev).eval()
R, S, V = a.rays, a.samples, a.views
q = torch.randn(R, S, 64, device=dev); feat = torch.randn(R, S, V, 64, device=dev)
rd = torch.randn(R, S, V, 4, device=dev); valid = torch.rand(R, S, V, device=dev) < 0.8
cnt = valid.sum(-1); empty = cnt == 0; valid = valid | empty[..., None]
layer = net.view_crosstrans[0]
def run():
    with torch.no_grad():
        return net._view_layer(layer, q, feat, rd, valid, cnt, bool(a.stats))
run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.iters): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / a.iters
mac_view = 4096 + 4096 + 4 * 8 + 8 * 64 + 64 * 8 + 8 * 64       # k, v, pos, attn per (group, view)
mac_grp = 4096 + 4096 + 2 * 64 * 256                               # q, out, FF per group
flop = 2.0 * R * S * (V * mac_view + mac_grp)
print(f"view layer: R={R} S={S} V={V} stats={a.stats}: {dt*1e3:.3f} ms  {flop/dt/1e12:.2f} TFLOP/s  ({flop/dt/157.3e12*100:.1f}% of fp32 MFMA peak)")

# ---- full GNT forward of one chunk (all layers; ray transformer etc. still on rocBLAS/torch)
pts = torch.randn(R, S, 3, device=dev); ray_d = torch.randn(R, 3, device=dev)
rgb_feat = torch.randn(R, S, V, 35, device=dev); mask = valid[..., None].float()
def full(stats):
    with torch.no_grad():
        return net(rgb_feat, rd, mask, pts, ray_d, ret_view_entropy=stats, ret_view_std=stats)
for stats in (False, True):
    full(stats); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): full(stats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    tot = 2.0 * R * S * (1048064 + 84416 * V) * a.depth / 8
    print(f"GNT forward depth={a.depth} stats={stats}: {dt*1e3:.2f} ms/chunk  {tot/dt/1e12:.2f} TFLOP/s; 1080p frame (1013 chunks) ~ {dt*1013:.1f} s")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    full(True); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
