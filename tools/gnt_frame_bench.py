#!/usr/bin/env python3
"""Whole PGDVSRenderer.forward with the GNT static renderer (random-init, 8 layers, 256 samples
per ray, chunk 1024 -- the reference's evaluator settings, configs/engine/evaluator_pgdvs.yaml)
on one synthetic target view at the NVIDIA-Dynamic-Scenes benchmark resolution (288 x 550,
10 spatial + 2 temporal source views; BASELINE.md section 1: the reference's only throughput
statement is "around 2 days on 8 A100" for 15 840 such images)."""
import argparse, pathlib, sys, time
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import numpy as np
import torch
from pgdvs_amd import synth
from pgdvs_amd.instantiate import load_config
from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=288)
ap.add_argument("--width", type=int, default=550)
ap.add_argument("--views", type=int, default=10)
ap.add_argument("--chunk", type=int, default=1024)
ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
dev = "cuda:0"
H, W, V = a.height, a.width, a.views
cfg = load_config(static_renderer="gnt")
rc = cfg.engine.engine_cfg.render_cfg
rc.chunk_size, rc.n_coarse_samples_per_ray = a.chunk, 256
rc.gnt_use_masked_spatial_src = False
rc.gnt_use_dyn_mask = True
torch.manual_seed(0)
model = PGDVSRenderer(cfg, render_cfg=rc).to(dev).eval()
video = synth.make_video(V, H, W, seed=3)
d = synth.to_torch(synth.make_view(video, V // 2, seed=1), dev)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
d["rgb_src_spatial"] = T(video["rgbs"])[None]
d["dyn_mask_src_spatial"] = T(video["dyn_masks"].astype(np.float32))[None, ..., None]
d["flat_cam_src_spatial"] = T(np.stack([synth.flat_cam(H, W, video["K3s"][i], video["c2ws"][i]) for i in range(V)]))[None]
d["depth_range"] = T(np.array([[0.8, 5.0]]))
def run():
    with torch.no_grad():
        return model.forward(d, render_cfg=rc, disable_tqdm=True)
ret = run(); torch.cuda.synchronize()
assert bool(torch.isfinite(ret["combined_rgb"]).all())
t = time.perf_counter()
for _ in range(a.iters): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / a.iters
print(f"PGDVSRenderer.forward, GNT static renderer, {H}x{W}, {V} spatial + 2 temporal views, 256 samples/ray, chunk {a.chunk}: "
      f"{dt:.3f} s per target view ({1/dt:.3f} frames/s)")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
