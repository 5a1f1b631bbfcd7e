#!/usr/bin/env python3
"""Round 4, review item 2 (a conservative depth bound before the rasteriser's fill pass): how many points / list entries such
a bound would drop, replayed on the CPU for the benchmark's clouds (rasteriser notes).
usage: python tools/raster_prune_sim.py [nominal|wide_baseline|noisy_depth]"""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/ml-pgdvs_amd']
import numpy as np
from scipy import ndimage
from oracle import oracle as orc
from pgdvs_amd import synth
scene = sys.argv[1] if len(sys.argv) > 1 else "nominal"
H, W, S, K = 1080, 1920, 24, 3
v = synth.make_video(S, H, W, seed=1234, scene=scene)
d = synth.make_view(v, 7, frac=0.4, seed=5)
cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
ndc = orc.points_to_ndc(cloud[:, :3], d["flat_cam_tgt"][0], H, W)
rng_x = 2.0 * W / H
# NDC -> pixel index (x reversed)
px = (W - 1) - ((ndc[:, 0] + rng_x / 2) * W - rng_x / 2) / rng_x
py = (H - 1) - ((ndc[:, 1] + 1.0) * H - 1.0) / 2.0
z = ndc[:, 2]
radius = 0.01
rpx = radius * W / rng_x
ok = (z >= 0) & (px > -rpx) & (px < W - 1 + rpx) & (py > -rpx) & (py < H - 1 + rpx)
print(scene, "points", len(z), "visible", ok.sum(), "radius px", rpx)
for C in (2, 3, 4):
    # cell of the point centre; all pixel centres of a C x C cell are within sqrt(2) * (C - 0.5) .. of any point in it
    diag = np.sqrt(2) * (C - 0.5)
    if diag >= rpx: continue
    cx, cy = np.floor((px + 0.5) / C).astype(int), np.floor((py + 0.5) / C).astype(int)
    ncx, ncy = -(-W // C), -(-H // C)
    inside = ok & (cx >= 0) & (cx < ncx) & (cy >= 0) & (cy < ncy)
    cid = cy[inside] * ncx + cx[inside]
    zz = z[inside]
    order = np.lexsort((zz, cid))
    cs, zs = cid[order], zz[order]
    first = np.r_[0, np.flatnonzero(np.diff(cs)) + 1]
    cnt = np.diff(np.r_[first, len(cs)])
    B = np.full(ncx * ncy, np.inf)
    has = cnt >= K
    B[cs[first[has]]] = zs[first[has] + K - 1]
    B = B.reshape(ncy, ncx)
    reach = int(np.ceil((rpx + C) / C))  # cells a disc can touch around its centre's cell
    Bmax = ndimage.maximum_filter(B, size=2 * reach + 1, mode="constant", cval=np.inf)
    # tile-level bound (what a per-tile list could use): max over the 16 x 16 tile's cells
    drop = np.zeros(len(z), bool)
    drop[np.flatnonzero(inside)] = zz > Bmax[cy[inside], cx[inside]]
    print(f"cell {C}: cells with >= K points {has.mean():.3f}; points droppable (per-point bound) {drop.sum() / ok.sum():.3f}")
    t = 16 // C if 16 % C == 0 else None
    if t:
        Bt = B.reshape(ncy // t if ncy % t == 0 else -1, t, -1, t) if (ncy % t == 0 and ncx % t == 0) else None
    # per-tile: an entry (point, tile) is droppable if z > max B over the tile's cells
    tx, ty = -(-W // 16), -(-H // 16)
    Bpad = np.full((ty * 16 // C + 1, tx * 16 // C + 1), -np.inf)
    Bpad[:ncy, :ncx] = B
    tb = np.full((ty, tx), -np.inf)
    per = 16 // C if 16 % C == 0 else None
    if per:
        tb = Bpad[:ty * per, :tx * per].reshape(ty, per, tx, per).max(axis=(1, 3))
        # entries: each point touches tiles overlapped by its disc box
        x0 = np.floor((px - rpx) / 16).astype(int); x1 = np.floor((px + rpx) / 16).astype(int)
        y0 = np.floor((py - rpx) / 16).astype(int); y1 = np.floor((py + rpx) / 16).astype(int)
        tot = kept = 0
        for dx in (0, 1):
            for dy in (0, 1):
                txi, tyi = x0 + dx, y0 + dy
                m = ok & (txi <= x1) & (tyi <= y1) & (txi >= 0) & (txi < tx) & (tyi >= 0) & (tyi < ty)
                tot += m.sum()
                kept += (z[m] <= tb[tyi[m], txi[m]]).sum()
        print(f"   per-tile bound: entries {tot} ({tot / ok.sum():.2f} per point), kept {kept / tot:.3f}")

# ---- the bound as built (csrc/raster.hip, raster_zmin / raster_bound kernels): per-pixel minimum depth of the point CENTRES
# (nearest pixel), per 4 x 4 block the K-th smallest of its 16 minima, per tile the maximum over its blocks
ix, iy = np.rint(px).astype(int), np.rint(py).astype(int)
m = ok & (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
zmin = np.full(H * W, np.inf, np.float32)
np.minimum.at(zmin, iy[m] * W + ix[m], z[m].astype(np.float32))
zb = np.full((-(-H // 4) * 4, -(-W // 4) * 4), np.inf, np.float32)
zb[:H, :W] = zmin.reshape(H, W)
blk = zb.reshape(zb.shape[0] // 4, 4, zb.shape[1] // 4, 4).transpose(0, 2, 1, 3).reshape(zb.shape[0] // 4, zb.shape[1] // 4, 16)
Bk = np.sort(blk, axis=2)[:, :, K - 1]
ty, tx = -(-H // 16), -(-W // 16)
Bp = np.full((ty * 4, tx * 4), -np.inf, np.float32)
# blocks that hold no pixel of the image do not constrain their tile
Bp[:Bk.shape[0], :Bk.shape[1]] = Bk
tb = Bp.reshape(ty, 4, tx, 4).max(axis=(1, 3))
x0 = np.floor((px - rpx) / 16).astype(int); x1 = np.floor((px + rpx) / 16).astype(int)
y0 = np.floor((py - rpx) / 16).astype(int); y1 = np.floor((py + rpx) / 16).astype(int)
tot = kept = 0
for dx in (0, 1):
    for dy in (0, 1):
        txi, tyi = x0 + dx, y0 + dy
        mm = ok & (txi <= x1) & (tyi <= y1) & (txi >= 0) & (txi < tx) & (tyi >= 0) & (tyi < ty)
        tot += mm.sum()
        kept += (z[mm] <= tb[tyi[mm], txi[mm]]).sum()
print(f"as built (K-th smallest of 16 per-pixel minima per 4 x 4 block, tile = max): kept {kept / tot:.3f} of {tot} entries; "
      f"tiles without a finite bound {np.mean(~np.isfinite(tb)):.4f}")
