"""Tuning aid: the synthetic video of pgdvs_amd.synth.make_video generated with torch on the GPU (seconds
instead of a minute at 1080p x 24).  Statistically the same scene (same cameras, surface, discs), NOT the same
bytes (different noise stream, float32 trigonometry) -- for kernel probes only, never for parity or bench."""
import numpy as np
import torch

from pgdvs_amd import synth


def make_video_gpu(S, H, W, dev="cuda:0", seed=1234, dyn_frac=0.15):
    g = torch.Generator(device=dev).manual_seed(seed)
    v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing="ij")
    rad = float(np.sqrt(dyn_frac * H * W / (2 * np.pi)))
    vel = np.array([6.0, -3.0]) * (W / 1920.0)
    c0 = np.array([[0.30 * W, 0.60 * H], [0.66 * W, 0.45 * H]])
    rgbs = torch.empty((S, H, W, 3), dtype=torch.float32, device=dev)
    depths = torch.empty((S, H, W), dtype=torch.float32, device=dev)
    masks = torch.empty((S, H, W), dtype=torch.bool, device=dev)
    K3s, c2ws = np.empty((S, 3, 3)), np.empty((S, 4, 4))
    surf = lambda x, y: 2.5 + 0.5 * torch.sin(1.3 * x) + 0.3 * torch.cos(1.7 * y)  # noqa: E731
    for i in range(S):
        K3, c2w = synth.frame_camera(i, S, H, W)
        K3s[i], c2ws[i] = K3, c2w
        M = torch.from_numpy(c2w[:3, :3] @ np.linalg.inv(K3)).to(dev)
        d = [M[k, 0] * u + M[k, 1] * v + M[k, 2] for k in range(3)]
        o = c2w[:3, 3]
        t = torch.full((H, W), 2.5, dtype=torch.float64, device=dev)
        for _ in range(6):
            t = (surf(o[0] + d[0] * t, o[1] + d[1] * t) - o[2]) / d[2]
        X, Y = o[0] + d[0] * t, o[1] + d[1] * t
        tex = torch.stack([0.5 + 0.4 * torch.sin(3.1 * X + 0.5) * torch.cos(2.3 * Y), 0.5 + 0.4 * torch.sin(2.7 * Y + 1.0),
                           0.5 + 0.4 * torch.cos(1.9 * X - 2.1 * Y)], -1)
        z = t
        m = torch.zeros((H, W), dtype=torch.bool, device=dev)
        for j in range(2):
            c = c0[j] + vel * i
            du, dv = (u - c[0]) / rad, (v - c[1]) / rad
            disc = du * du + dv * dv < 1.0
            m |= disc
            z = torch.where(disc, 1.0 + 0.08 * du + 0.05 * dv + 0.2 * j, z)
            obj = torch.stack([0.6 + 0.3 * torch.sin(4 * du + j), 0.4 + 0.3 * torch.cos(3 * dv), 0.5 + 0.3 * torch.sin(5 * du * dv + 1)], -1)
            tex = torch.where(disc[..., None], obj, tex)
        rgbs[i] = (tex + 0.01 * torch.randn(tex.shape, device=dev, generator=g, dtype=torch.float64)).clamp(0, 1).float()
        depths[i] = z.float()
        masks[i] = m
    return dict(rgbs=rgbs, depths=depths, dyn_masks=masks, K3s=K3s, c2ws=c2ws, vel=vel, rad=rad)


def bench_cloud(S=24, H=1080, W=1920, dev="cuda:0"):
    """(cloud[n,6], video dict, flat target camera) of the benchmark-like scene"""
    from pgdvs_amd import ops

    v = make_video_gpu(S, H, W, dev)
    cloud, cnt = ops.static_aggregate(v["rgbs"], v["depths"], v["dyn_masks"].view(torch.uint8), v["K3s"], v["c2ws"], capacity=S * H * W)
    n = ops.checked_count(cnt, "agg")
    i, frac = 7, 0.4
    fr = (i + frac) / max(S - 1, 1)
    K1, K2 = v["K3s"][i], v["K3s"][i + 1]
    ct = synth._pose(2.0 * (fr - 0.5) + 0.3, 0.6 * (fr - 0.5) - 0.2, [0.02 * (i + frac), 0.003 * (i + frac) + 0.004, -0.01])
    fc = synth.flat_cam(H, W, K1 * (1 - frac) + K2 * frac, ct)
    return cloud[:n], v, fc
