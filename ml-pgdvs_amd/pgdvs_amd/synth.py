"""Synthetic (depth, flow, pose, RGB) inputs in the reference's ``data``-dict layout
(key list: pgdvs/datasets/nvidia_eval.py:545-603).

There is no network access for the real datasets, so benchmarks, smoke and parity tests
use this seeded generator.  The scene is geometrically consistent across frames (a
height-field background seen by a slowly moving camera plus a moving foreground disc)
so that the static-aggregation dedup, the flow warp and the softsplat metric behave as
they do on real sequences: later frames are mostly covered by the accumulated cloud,
and colour-consistent flows get softsplat weights near 1.
"""
from __future__ import annotations

import numpy as np


def _pose(yaw_deg, pitch_deg, t):
    y, p = np.deg2rad(yaw_deg), np.deg2rad(pitch_deg)
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    c2w = np.eye(4)
    c2w[:3, :3] = Ry @ Rx
    c2w[:3, 3] = t
    return c2w


def _surface(x, y):
    return 2.5 + 0.5 * np.sin(1.3 * x) + 0.3 * np.cos(1.7 * y)


def _texture(x, y):
    return np.stack(
        [0.5 + 0.4 * np.sin(3.1 * x + 0.5) * np.cos(2.3 * y), 0.5 + 0.4 * np.sin(2.7 * y + 1.0),
         0.5 + 0.4 * np.cos(1.9 * x - 2.1 * y)], axis=-1)


def _background(H, W, K3, c2w):
    """Ray / height-field intersection by fixed-point iteration -> z-depth[H,W], rgb[H,W,3]."""
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    pix = np.stack([u, v, np.ones_like(u)], -1)
    d = pix @ np.linalg.inv(K3).T @ c2w[:3, :3].T  # [H,W,3], camera-frame z == 1
    o = c2w[:3, 3]
    t = np.full((H, W), 2.5)
    for _ in range(6):
        X = o[0] + d[..., 0] * t
        Y = o[1] + d[..., 1] * t
        t = (_surface(X, Y) - o[2]) / d[..., 2]
    X = o[0] + d[..., 0] * t
    Y = o[1] + d[..., 1] * t
    return t, _texture(X, Y)


RIG_CAMERAS = 12  # the NVIDIA Dynamic Scenes rig (12 cameras; the monocular protocol takes camera i % 12 at time i)


def frame_camera(i, S, H, W, scene="nominal"):
    """Source camera i of S: small yaw sweep + translation along x.  ``scene="wide_baseline"``: the camera
    of frame i is camera i % 12 of a fixed 12-camera rig spanning the same baseline (pgdvs/datasets/nvidia_eval.py's
    monocular protocol): consecutive frames are a rig spacing apart and frame 12 jumps back across the whole rig."""
    f = 0.9 * W
    K3 = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    if scene == "wide_baseline":
        c = i % RIG_CAMERAS
        fr = c / (RIG_CAMERAS - 1.0)
        return K3, _pose(2.0 * (fr - 0.5), 0.6 * (fr - 0.5), [0.46 * fr, 0.069 * fr, 0.0])
    frac = i / max(S - 1, 1)
    c2w = _pose(2.0 * (frac - 0.5), 0.6 * (frac - 0.5), [0.02 * i, 0.003 * i, 0.0])
    return K3, c2w


def make_video(S, H, W, seed=1234, dyn_frac=0.15, scene="nominal"):
    """S source frames: dict(rgbs[S,H,W,3] f32, depths[S,H,W] f32, dyn_masks[S,H,W] bool,
    K3s[S,3,3] f64, c2ws[S,4,4] f64, centers[S,2,2]).
    ``scene``: "nominal" (smooth camera path, exact depth), "wide_baseline" (12-camera rig cycled per frame, see
    ``frame_camera``), "noisy_depth" (estimated-depth statistics: 1.5 % multiplicative noise on every depth and flying
    pixels -- depths between foreground and background -- in a 2-pixel band along the objects' borders)."""
    assert scene in ("nominal", "wide_baseline", "noisy_depth"), scene
    rng = np.random.default_rng(seed)
    rng_depth = np.random.default_rng(seed + 99991)  # (its own stream: the nominal scene's draws are unchanged)
    rgbs = np.empty((S, H, W, 3), np.float32)
    depths = np.empty((S, H, W), np.float32)
    masks = np.empty((S, H, W), bool)
    K3s = np.empty((S, 3, 3))
    c2ws = np.empty((S, 4, 4))
    # two discs covering ~dyn_frac of the image, moving (6,-3) px/frame at 1080p
    rad = np.sqrt(dyn_frac * H * W / (2 * np.pi))
    vel = np.array([6.0, -3.0]) * (W / 1920.0)
    c0 = np.array([[0.30 * W, 0.60 * H], [0.66 * W, 0.45 * H]])
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    centers = np.empty((S, 2, 2))

    def frame(i):
        K3, c2w = frame_camera(i, S, H, W, scene)
        z, tex = _background(H, W, K3, c2w)
        z_bg = z
        m = np.zeros((H, W), bool)
        border = np.zeros((H, W), bool)
        cs = np.empty((2, 2))
        for j in range(2):
            c = c0[j] + vel * i
            cs[j] = c
            du, dv = (u - c[0]) / rad, (v - c[1]) / rad
            disc = du * du + dv * dv < 1.0
            m |= disc
            border |= np.abs(np.sqrt(du * du + dv * dv) - 1.0) * rad < 2.0
            z = np.where(disc, 1.0 + 0.08 * du + 0.05 * dv + 0.2 * j, z)
            obj = np.stack([0.6 + 0.3 * np.sin(4 * du + j), 0.4 + 0.3 * np.cos(3 * dv), 0.5 + 0.3 * np.sin(5 * du * dv + 1)], -1)
            tex = np.where(disc[..., None], obj, tex)
        return K3, c2w, z.astype(np.float32), tex, m, cs, border, z_bg.astype(np.float32)

    # frames are independent except for the noise stream, which is drawn in frame order: the geometry
    # (numpy ufuncs release the GIL) runs on a thread pool, a few frames ahead of the sequential part
    import concurrent.futures
    import os

    workers = max(1, min(S, (os.cpu_count() or 1), 16))
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        pending = {}
        nxt = 0
        for i in range(S):
            while nxt < S and nxt < i + 2 * workers:
                pending[nxt] = ex.submit(frame, nxt)
                nxt += 1
            K3s[i], c2ws[i], depths[i], tex, masks[i], centers[i], border, z_bg = pending.pop(i).result()
            rgbs[i] = np.clip(tex + rng.normal(0, 0.01, tex.shape), 0, 1).astype(np.float32)
            if scene == "noisy_depth":
                # flying pixels: a depth somewhere between the object (~1.1) and the background behind it
                mix = rng_depth.random((H, W)).astype(np.float32)
                depths[i] = np.where(border, np.float32(1.1) + mix * (z_bg - np.float32(1.1)), depths[i])
                depths[i] *= (1.0 + 0.015 * rng_depth.standard_normal((H, W))).astype(np.float32)
    return dict(rgbs=rgbs, depths=depths, dyn_masks=masks, K3s=K3s, c2ws=c2ws, centers=centers, vel=vel, rad=rad)


def flat_cam(H, W, K3, c2w):
    K4 = np.eye(4)
    K4[:3, :3] = K3
    return np.concatenate(([H, W], K4.reshape(-1), np.asarray(c2w).reshape(-1))).astype(np.float32)


def make_view(video, i, frac=0.4, seed=0, with_noise=True):
    """The reference's data dict (batch 1) for a target view between frames i and i+1 at
    time i+frac, with the two temporally closest frames as ``*_src_temporal``."""
    S, H, W = video["depths"].shape
    j = min(i + 1, S - 1)
    rng = np.random.default_rng(seed + 7919 * i)
    K1, c1 = video["K3s"][i], video["c2ws"][i]
    K2, c2 = video["K3s"][j], video["c2ws"][j]
    # rigid background flow 1->2 from depth; object flow = disc velocity + N(0,1)*scale
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    pix = np.stack([u, v, np.ones_like(u)], -1)
    d = pix @ np.linalg.inv(K1).T @ c1[:3, :3].T
    X = c1[:3, 3] + d * video["depths"][i][..., None].astype(np.float64)
    w2c2 = np.linalg.inv(c2)
    Xc = X @ w2c2[:3, :3].T + w2c2[:3, 3]
    p = Xc @ K2.T
    uv2 = p[..., :2] / p[..., 2:3]
    flow = uv2 - pix[..., :2]
    m = video["dyn_masks"][i]
    obj_flow = video["vel"] * (j - i) + rng.normal(0, 1.0 * W / 1920.0, (H, W, 2))
    flow = np.where(m[..., None], obj_flow, flow).astype(np.float32)
    # target camera: interpolate the pose parameters, nudge off the source trajectory
    Kt = K1 * (1 - frac) + K2 * frac
    ct = _pose(0.0, 0.0, [0, 0, 0])
    fr = (i + frac) / max(S - 1, 1)
    ct = _pose(2.0 * (fr - 0.5) + 0.3, 0.6 * (fr - 0.5) - 0.2, [0.02 * (i + frac), 0.003 * (i + frac) + 0.004, -0.01])
    data = {
        "rgb_src_temporal": np.stack([video["rgbs"][i], video["rgbs"][j]])[None],
        "depth_src_temporal": np.stack([video["depths"][i], video["depths"][j]])[None, ..., None],
        "dyn_mask_src_temporal": np.stack([video["dyn_masks"][i], video["dyn_masks"][j]])[None, ..., None].astype(np.float32),
        "flow_fwd": flow[None],
        "flow_fwd_occ_mask": np.zeros((1, H, W, 1), np.float32),
        "flat_cam_tgt": flat_cam(H, W, Kt, ct)[None],
        "flat_cam_src_temporal": np.stack([flat_cam(H, W, K1, c1), flat_cam(H, W, K2, c2)])[None],
        "time_tgt": np.array([[i + frac]], np.float32),
        "time_src_temporal": np.array([[i, j]], np.float32),
    }
    if with_noise:
        data["static_noise"] = rng.standard_normal((1, 3, H, W)).astype(np.float32)
    return data


def add_track_window(data, video, i, n_side=2, seed=0, step=3, p_closest_visible=0.5):
    """Adds the tracker-window keys (pgdvs_renderer_dyn_track.py:599-716: ``*_src_temporal_track_fwd2tgt`` /
    ``_bwd2tgt``, ``n_actual_*``) around the view of make_view(video, i) and synthetic point
    tracks standing in for a tracker's output: every ``step``-th dynamic pixel of the track
    frames, moved with the disc velocity, with random visibility drop-outs.
    Frame order of the tracks: [fwd2tgt..., i, i+1, bwd2tgt...]."""
    S, H, W = video["depths"].shape
    j = min(i + 1, S - 1)
    rng = np.random.default_rng(seed + 104729 * i)
    fwd = [f for f in range(i - n_side, i) if f >= 0]
    bwd = [f for f in range(j + 1, j + 1 + n_side) if f < S]
    closest = [i, j]

    def pack(frames, n):
        def pad(a):
            a = np.asarray(a)
            return np.concatenate([a, np.zeros((n - a.shape[0],) + a.shape[1:], a.dtype)], 0)[None]
        fr = list(frames)
        return {
            "rgb": pad(video["rgbs"][fr].reshape(len(fr), H, W, 3)), "depth": pad(video["depths"][fr].reshape(len(fr), H, W)[..., None]),
            "dyn_mask": pad(video["dyn_masks"][fr].reshape(len(fr), H, W)[..., None].astype(np.float32)),
            "flat_cam": pad(np.stack([flat_cam(H, W, video["K3s"][f], video["c2ws"][f]) for f in fr]).reshape(len(fr), 34)
                            if fr else np.zeros((0, 34), np.float32)),
            "time": pad(np.array(fr, np.float32))}

    for suffix, frames in (("_track_fwd2tgt", fwd), ("_track_bwd2tgt", bwd)):
        for k, v in pack(frames, n_side).items():
            data[f"{k}_src_temporal{suffix}"] = v
    data["n_actual_temporal_track_fwd2tgt"] = np.array([[len(fwd)]], np.int64)
    data["n_actual_temporal_track_bwd2tgt"] = np.array([[len(bwd)]], np.int64)
    data["n_actual_temporal"] = np.array([[2]], np.int64)

    window = fwd + closest + bwd
    tracks, vis = [], []
    for q in fwd + bwd:
        rows, cols = np.nonzero(video["dyn_masks"][q])
        rows, cols = rows[::step], cols[::step]
        pos = np.stack([cols, rows], -1).astype(np.float64)  # (col,row)
        tr = np.stack([pos + video["vel"] * (f - q) + rng.normal(0, 0.05, pos.shape) for f in window], 1)
        inside = (tr[..., 0] >= 0) & (tr[..., 0] <= W - 1) & (tr[..., 1] >= 0) & (tr[..., 1] <= H - 1)
        v = inside & (rng.random(inside.shape) < 0.85)
        v[:, len(fwd):len(fwd) + 2] &= (rng.random((pos.shape[0], 1)) < p_closest_visible)
        tracks.append(tr.astype(np.float32))
        vis.append(v)
    N = len(window)
    data["track_tracks"] = [np.concatenate(tracks) if tracks else np.zeros((0, N, 2), np.float32)]
    data["track_visibles"] = [np.concatenate(vis) if vis else np.zeros((0, N), bool)]
    return data


def to_torch(d, device):
    import torch

    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.dtype != np.float64:
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
        elif isinstance(v, list) and v and all(isinstance(e, np.ndarray) for e in v):
            out[k] = [torch.from_numpy(np.ascontiguousarray(e)).to(device) for e in v]
        else:
            out[k] = v
    return out


DEFAULT_RENDER_CFG = dict(
    render_stride=1, chunk_size=1024, sample_inv_uniform=True, n_coarse_samples_per_ray=256, n_fine_samples_per_ray=0,
    pure_gnt=False, pure_gnt_with_dyn_mask=False, gnt_use_dyn_mask=False, gnt_use_masked_spatial_src=True,
    mask_oob_n_proj_thres=1, mask_invalid_n_proj_thres=4, st_pcl_remove_outlier=False, st_pcl_outlier_knn=50,
    st_pcl_outlier_std_thres=0.1, st_render_pcl_pt_radius=0.01, st_render_pcl_pts_per_pixel=1,
    dyn_pcl_remove_outlier=False, dyn_pcl_outlier_knn=50, dyn_pcl_outlier_std_thres=0.1, dyn_render_type="softsplat",
    dyn_render_pcl_pt_radius=0.01, dyn_render_pcl_pts_per_pixel=1, dyn_render_track_temporal="none",
    dyn_pcl_track_track2base_thres_mult=50, dyn_render_use_flow_consistency=False,
)
