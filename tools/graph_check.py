#!/usr/bin/env python3
"""Diagnostic: every op of the point-renderer path captured into a HIP graph on its own and replayed; the replays must
reproduce the eager results bit for bit (run on the GPU box)."""
import pathlib
import sys

R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
sys.path.insert(0, str(R0 / "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pgdvs_amd import ops, synth  # noqa: E402

dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
H, W, S = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (270, 480, 6)
v = synth.make_video(S, H, W, seed=3)
rgbs, depths, masks = T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]).view(torch.uint8)
d = synth.make_view(v, 1, seed=3)
cam = ops.cam_prep(T(d["flat_cam_tgt"][0]))
cap = S * H * W


import os  # noqa: E402


def agg():
    if os.environ.get("GC_NOXYZ"):
        c, k = ops.static_aggregate(rgbs, depths, masks, v["K3s"], v["c2ws"], capacity=cap)
        return c, k, c[:, :3].contiguous()
    return ops.static_aggregate(rgbs, depths, masks, v["K3s"], v["c2ws"], capacity=cap, return_xyz=True)


cloud_e, cnt_e, xyz_e = agg()
torch.cuda.synchronize()
n = int(cnt_e.item())


def raster(cloud, cnt, xyz):
    r = ops.points_raster(xyz, cloud[:, 3:], cam, 0.01, 3, H, W, n_points_dev=cnt, want_fragments=True)
    return r["idx"], r["rgb"], r["mask"], r["zbuf"], r["dist2"]


print("eager aggregation done", n, flush=True)
ras_e = raster(cloud_e, cnt_e, xyz_e)
torch.cuda.synchronize()
print("eager raster done", flush=True)


def knn(xyz, cnt):
    return ops.knn_mean_dist(xyz, cnt.to(torch.int32), 50, algo=2)


kn_e = knn(xyz_e[: max(n, 1)].contiguous(), cnt_e)
torch.cuda.synchronize()
print("eager knn done", flush=True)


def check(name, fn, ref, same):
    """capture fn through pgdvs_amd.runtime.GraphedRender (the harness bench.py --launch graph uses), replay four times"""
    if os.environ.get("GC_OPS") and name not in os.environ["GC_OPS"].split(","):
        return
    from pgdvs_amd.runtime import GraphedRender

    gr = GraphedRender(lambda d: fn(), {"dummy": torch.zeros(4, device=dev)})
    bad = 0
    for it in range(4):
        junk = torch.full((16 << 20,), float(it + 1), device=dev)  # other allocator traffic between replays
        del junk
        out = gr({"dummy": torch.zeros(4, device=dev)})
        gr.stream.synchronize()
        bad += not same(out, ref)
    print(f"{name}: {'ok' if bad == 0 else f'MISMATCH in {bad} of 4 replays'}", flush=True)


fc = T(d["flat_cam_tgt"][0])
check("cam_prep", lambda: ops.cam_prep(fc), cam, torch.equal)
check("points_raster", lambda: raster(cloud_e, cnt_e, xyz_e), ras_e, lambda o, r: all(torch.equal(a, b) for a, b in zip(o, r)))
check("knn_mean_dist", lambda: knn(xyz_e[: max(n, 1)].contiguous(), cnt_e), kn_e, lambda o, r: torch.equal(o[:n], r[:n]))

check("static_aggregate", agg, (cloud_e, cnt_e, xyz_e),
      lambda o, r: int(o[1].item()) == n and torch.equal(o[0][:n], r[0][:n]) and torch.equal(o[2][:n], r[2][:n]))
