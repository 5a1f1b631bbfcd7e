#!/usr/bin/env python3
"""Tuning aid: what would raster_tile's work counters be for other visiting orders of a tile's list?
Emulates the kernel's loop (64-entry chunks, quadrant box + hierarchical-z cull, per-pixel test, top-K insert)
on a sample of tiles of the benchmark-like scene for: the cloud order (today), z-bucketed orders of several
bucket widths (relative width 2^-m: bucket = float bits >> (23 - m)), and the exact z order.
Prints per-wave: staged entries (until early exit when the order allows one), tests, insert events, lane inserts."""
import pathlib
import sys

import numpy as np

R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
sys.path.insert(0, str(R0 / "tools"))
import torch  # noqa: E402

from fast_video import bench_cloud  # noqa: E402
from pgdvs_amd import ops  # noqa: E402

K, RADIUS, H, W = 3, 0.01, 1080, 1920
cloud, v, fc = bench_cloud()
# pixel-index coordinates of all points (pixel i centred at i, as pytorch3d's +0.5 centres in OpenCV pixel
# units) and view depth: plain pinhole projection in float64 -- statistics only, no parity claim
fcn = np.asarray(fc, np.float64)
K4, c2w = fcn[2:18].reshape(4, 4), fcn[18:34].reshape(4, 4)
w2c = np.linalg.inv(c2w)
Xc = cloud[:, :3].cpu().numpy().astype(np.float64) @ w2c[:3, :3].T + w2c[:3, 3]
z = Xc[:, 2]
px = K4[0, 0] * Xc[:, 0] / z + K4[0, 2] - 0.5
py = K4[1, 1] * Xc[:, 1] / z + K4[1, 2] - 0.5
s = min(H, W) / 2.0
rpx = RADIUS * s
rng = np.random.default_rng(0)
tiles = [(int(rng.integers(4, H // 16 - 4)), int(rng.integers(4, W // 16 - 4))) for _ in range(40)]


def simulate(order_fn, early_exit):
    tot = np.zeros(5)
    for ty, tx in tiles:
        x0, y0 = tx * 16, ty * 16
        sel = np.nonzero((px > x0 - rpx - 1) & (px < x0 + 16 + rpx) & (py > y0 - rpx - 1) & (py < y0 + 16 + rpx) & (z >= 0))[0]
        order, zlo = order_fn(sel)
        sel = sel[order]
        for q in range(4):
            qx0, qy0 = x0 + (q & 1) * 8, y0 + (q >> 1) * 8
            gx, gy = np.meshgrid(np.arange(qx0, qx0 + 8), np.arange(qy0, qy0 + 8))
            gx, gy = gx.reshape(-1).astype(np.float64), gy.reshape(-1).astype(np.float64)
            keyz = np.full((64, K), np.inf)
            keyi = np.full((64, K), 1 << 40, dtype=np.int64)
            staged = tests = events = lanes = 0
            for c0 in range(0, len(sel), 64):
                ch = sel[c0:c0 + 64]
                zc = keyz[:, K - 1].max()
                if early_exit and zlo is not None and zlo[c0] > zc:
                    break
                staged += len(ch)
                keep = ch[(px[ch] >= qx0 - rpx) & (px[ch] <= qx0 + 7 + rpx) & (py[ch] >= qy0 - rpx) & (py[ch] <= qy0 + 7 + rpx) & (z[ch] <= zc)]
                for p in keep:
                    tests += 1
                    d2 = (gx - px[p]) ** 2 + (gy - py[p]) ** 2
                    ok = (d2 < rpx * rpx) & ((z[p] < keyz[:, K - 1]) | ((z[p] == keyz[:, K - 1]) & (p < keyi[:, K - 1])))
                    n = int(ok.sum())
                    if n:
                        events += 1
                        lanes += n
                        for lane in np.nonzero(ok)[0]:
                            kz, ki = list(keyz[lane]), list(keyi[lane])
                            kz.append(z[p]); ki.append(p)
                            o = sorted(range(K + 1), key=lambda t: (kz[t], ki[t]))[:K]
                            keyz[lane] = [kz[t] for t in o]; keyi[lane] = [ki[t] for t in o]
            tot += [1, staged, tests, events, lanes]
    return tot[1:] / tot[0]


def cloud_order(sel):
    return np.arange(len(sel)), None


def bucket_order(m):
    def f(sel):
        b = z[sel].astype(np.float32).view(np.uint32) >> (23 - m)
        o = np.argsort(b, kind="stable")
        lo = ((b[o].astype(np.uint32)) << (23 - m)).view(np.float32)
        return o, lo
    return f


def exact_order(sel):
    o = np.argsort(z[sel], kind="stable")
    return o, z[sel][o]


print("order                 staged   tests  events  lane-inserts   (per wave, %d tiles)" % len(tiles))
for name, fn, ee in [("cloud order (today)", cloud_order, False)] + [(f"z buckets 2^-{m}", bucket_order(m), True) for m in (6, 8, 10, 12)] + [
        ("exact z order", exact_order, True)]:
    r = simulate(fn, ee)
    print(f"{name:20s} {r[0]:8.0f} {r[1]:7.0f} {r[2]:7.0f} {r[3]:9.0f}", flush=True)
