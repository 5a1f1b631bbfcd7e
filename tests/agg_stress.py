"""Adversarial sweep for the static aggregation's fp32 screening (run on the GPU box: ``python tests/agg_stress.py [n]``).

Small random scenes whose cameras differ by large rotations and translations, with depths over four decades, so that
many projections fall far outside the later frames, behind their cameras or next to their image planes -- the cases the
error bound of ``screen_frames`` (csrc/static_agg.hip) has to get right without the smooth orbits of the benchmark
video.  Every cloud must equal the oracle's (``_compute_pcl_proj_mask`` in fp64, nvidia_eval_pure_geo.py:257-277) bit
for bit (run it once more with PGDVS_AGG_ORDERED=1 for the ordered chain; the library reads the switch once per process).
Test infrastructure (the oracle is the checker): tests/test_gpu_round3.py runs a 40-scene slice."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-pgdvs_amd"))

from oracle import oracle as orc  # noqa: E402  (checker only)
from pgdvs_amd import ops  # noqa: E402


def rot(rng, max_angle):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def scene(seed):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(2, 7))
    H, W = int(rng.integers(6, 70)), int(rng.integers(6, 90))
    kind = seed % 4
    rgbs = rng.random((S, H, W, 3), dtype=np.float32)
    lo, hi = [(0.5, 4.0), (0.01, 100.0), (1.0, 1.0001), (0.05, 50.0)][kind]
    depths = np.exp(rng.uniform(np.log(lo), np.log(hi), (S, H, W))).astype(np.float32)
    if kind == 3:  # planes: many exact ties and integer-valued projections
        depths[:] = np.float32(2.0)
    masks = rng.random((S, H, W)) < rng.uniform(0.0, 0.5)
    K3s = np.empty((S, 3, 3))
    c2ws = np.empty((S, 4, 4))
    for i in range(S):
        f = rng.uniform(0.3, 3.0) * W
        K3s[i] = [[f, 0, rng.uniform(0.3, 0.7) * W], [0, f * rng.uniform(0.8, 1.25), rng.uniform(0.3, 0.7) * H], [0, 0, 1]]
        if kind == 3:
            K3s[i] = [[float(W), 0, W / 2], [0, float(W), H / 2], [0, 0, 1]]
        c2w = np.eye(4)
        c2w[:3, :3] = rot(rng, [0.3, 3.1, 0.02, 0.0][kind])
        c2w[:3, 3] = rng.normal(0, [0.3, 2.0, 0.01, 0.0][kind], 3)
        if kind == 3:
            c2w[:3, 3] = [float(rng.integers(-3, 4)) * 2.0 / W, float(rng.integers(-3, 4)) * 2.0 / W, 0.0]
        c2ws[i] = c2w
    return rgbs, depths, masks, K3s, c2ws


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = "cuda:0"
    bad = 0
    total = 0
    for seed in range(n):
        rgbs, depths, masks, K3s, c2ws = scene(seed)
        want = orc.aggregate_static_pcl(rgbs, depths, masks, K3s, c2ws).astype(np.float32)
        cloud, cnt = ops.static_aggregate(torch.from_numpy(rgbs).to(dev), torch.from_numpy(depths).to(dev),
                                          torch.from_numpy(masks).to(dev), K3s, c2ws)
        k = ops.checked_count(cnt, "pgdvs_static_aggregate")
        got = cloud[:k].cpu().numpy()
        if not (got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))):
            bad += 1
            print(f"seed {seed} (kind {seed % 4}, S={depths.shape[0]}, {depths.shape[1]}x{depths.shape[2]}): "
                  f"{k} points against the oracle's {want.shape[0]}")
        total += want.shape[0]
    print(f"{n} scenes, {total} points in all, {bad} mismatching clouds (PGDVS_AGG_ORDERED={os.environ.get('PGDVS_AGG_ORDERED', '0')})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
