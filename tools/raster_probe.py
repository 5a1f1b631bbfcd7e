#!/usr/bin/env python3
"""Tuning aid: time pgdvs_points_raster on the benchmark-like cloud (tools/fast_video.py) with HIP events
per kernel; PGDVS_RASTER_STATS=1 prints the tile kernel's work counters."""
import ctypes
import pathlib
import sys
import time

R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
sys.path.insert(0, str(R0 / "tools"))
import torch  # noqa: E402

from fast_video import bench_cloud  # noqa: E402
from pgdvs_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
t0 = time.time()
cloud, v, fc = bench_cloud()
torch.cuda.synchronize()
print(f"cloud {cloud.shape[0]} points in {time.time() - t0:.1f} s", flush=True)
cam = ops.cam_prep(torch.from_numpy(fc).to(dev))
H, W, K = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = _lib.load()
for _ in range(3):
    r = ops.points_raster(cloud, cloud[:, 3:], cam, 0.01, K, H, W, rgb_planar=True)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.pgdvs_prof_enable(1)
lib.pgdvs_prof_report(buf, len(buf))
n = 10
for _ in range(n):
    r = ops.points_raster(cloud, cloud[:, 3:], cam, 0.01, K, H, W, rgb_planar=True)
torch.cuda.synchronize()
lib.pgdvs_prof_report(buf, len(buf))
lib.pgdvs_prof_enable(0)
for line in buf.value.decode().strip().splitlines():
    name, calls, total = line.split()
    print(f"{name:28s} {float(total) / int(calls) * 1e3:9.1f} us")
print("checksum", float(r["rgb"].double().sum()), float(r["mask"].sum()))
