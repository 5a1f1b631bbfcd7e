#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep over the kernels with data-dependent fast paths (kNN
thread-per-query pass, rasteriser depth cull, splat flag pre-pass): many seeds, ragged sizes.
Diagnostic (run on the GPU box); the fixed cases live in tests/test_gpu_parity.py."""
import os, pathlib, sys
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import numpy as np
import torch
from oracle import oracle as orc
from pgdvs_amd import ops, synth
from pgdvs_amd.instantiate import load_config
from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
N = lambda t: t.detach().cpu().numpy()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rng = np.random.default_rng(1000 + seed)
    # ---- kNN: surface + clusters + duplicates, K in the thread-per-query set and outside it
    n = int(rng.integers(60, 6000)); K = int(rng.choice([4, 8, 16, 20, 50, 7, 30]))
    u = rng.uniform(-1, 1, (n, 2)); pts = np.stack([u[:, 0], u[:, 1], 2 + 0.3 * np.sin(3 * u[:, 0]) + rng.normal(0, 0.003, n)], 1)
    pts[: n // 20] += rng.normal(0, 0.7, (n // 20, 3)); pts[n // 2: n // 2 + n // 30] = pts[:n // 30]
    pts = pts.astype(np.float32)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    a = N(ops.knn_mean_dist(T(pts), cnt, K, algo=2))[:n]
    ref = orc.knn_mean_dist(pts, K)
    ok = np.array_equal(a.view(np.uint32), ref.view(np.uint32)); bad += not ok
    print(f"seed {seed}: knn n={n} K={K} {'ok' if ok else 'MISMATCH'}")
    # ---- rasteriser: layered depths, random radius / K
    H, W = int(rng.integers(20, 70)), int(rng.integers(20, 70)); m = int(rng.integers(500, 20000)); Kp = int(rng.integers(1, 9))
    radius = float(rng.uniform(0.01, 0.09))
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    z = 1.0 + 0.2 * rng.integers(0, 5, m) + np.where(rng.random(m) < 0.4, 0.0, rng.uniform(0, 0.05, m))
    xy = rng.uniform(-1.3, 1.3, (m, 2)) * z[:, None] * 0.6
    p3 = np.concatenate([xy, z[:, None]], 1).astype(np.float32); rgb = rng.random((m, 3), dtype=np.float32)
    cloud = T(np.concatenate([p3, rgb], 1))
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, Kp, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(p3, fc, H, W, radius, Kp)
    ok = np.array_equal(N(r["idx"]), idx) and np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32)); bad += not ok
    print(f"seed {seed}: raster {H}x{W} n={m} K={Kp} r={radius:.3f} {'ok' if ok else 'MISMATCH'}")
    # ---- whole view (splat with the flag pre-pass, filter on), small
    Hh, Ww, S = int(rng.integers(40, 90)), int(rng.integers(48, 120)), 3
    v = synth.make_video(S, Hh, Ww, seed=seed); d = synth.make_view(v, int(rng.integers(0, 2)), seed=seed)
    cfg = load_config(static_renderer="geo"); rc = cfg.engine.engine_cfg.render_cfg
    rc.dyn_pcl_remove_outlier, rc.dyn_pcl_outlier_knn, rc.st_render_pcl_pts_per_pixel = True, 20, 3
    model = PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(DEV).eval()
    cloud_s = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    data = synth.to_torch(d, DEV); data["st_pcl_rgb"] = T(cloud_s.astype(np.float32))[None]
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d); od["st_pcl_rgb"] = cloud_s[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    err = float(np.abs(N(ret["combined_rgb"]) - o["combined_rgb"]).max())
    ok = err < 1e-4 and np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"]); bad += not ok
    print(f"seed {seed}: view {Hh}x{Ww} max|d|={err:.2e} {'ok' if ok else 'MISMATCH'}")
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
