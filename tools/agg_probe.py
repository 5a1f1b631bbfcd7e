#!/usr/bin/env python3
"""Tuning aid: time pgdvs_static_aggregate on the benchmark-like video (tools/fast_video.py) per kernel."""
import ctypes
import pathlib
import sys
import time

R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
sys.path.insert(0, str(R0 / "tools"))
import torch  # noqa: E402

from fast_video import make_video_gpu  # noqa: E402
from pgdvs_amd import _lib, ops  # noqa: E402

S, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), 1080, 1920
v = make_video_gpu(S, H, W)
m8 = v["dyn_masks"].view(torch.uint8)
lib = _lib.load()
for _ in range(3):
    cloud, cnt = ops.static_aggregate(v["rgbs"], v["depths"], m8, v["K3s"], v["c2ws"], capacity=S * H * W)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    cloud, cnt = ops.static_aggregate(v["rgbs"], v["depths"], m8, v["K3s"], v["c2ws"], capacity=S * H * W)
torch.cuda.synchronize()
print(f"whole call: {(time.perf_counter() - t0) / n * 1e3:.3f} ms, {int(cnt)} points")
buf = ctypes.create_string_buffer(1 << 16)
lib.pgdvs_prof_enable(1)
lib.pgdvs_prof_report(buf, len(buf))
for _ in range(n):
    ops.static_aggregate(v["rgbs"], v["depths"], m8, v["K3s"], v["c2ws"], capacity=S * H * W)
torch.cuda.synchronize()
lib.pgdvs_prof_report(buf, len(buf))
lib.pgdvs_prof_enable(0)
tot = 0.0
for line in buf.value.decode().strip().splitlines():
    name, calls, total = line.split()
    tot += float(total) / n
    print(f"{name:20s} {int(calls) // n:4d} x {float(total) / int(calls) * 1e3:9.1f} us = {float(total) / n * 1e3:9.1f} us per call")
print(f"sum of kernels {tot * 1e3:.1f} us")
