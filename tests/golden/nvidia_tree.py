"""Deterministic synthetic dataset tree in the on-disk layout the reference's NVIDIA Dynamic
Scenes evaluation datasets read (pgdvs/datasets/nvidia_eval.py:180-225,608-645,878-1011,
nvidia_eval_pure_geo.py:183-200).  Shared by the golden generator (which points the REFERENCE
dataset classes at it) and by tests/test_host_cpu.py (which points the mirror at a freshly
rebuilt copy): the files are data, the layout is the reference's.

  raw/<scene>/dense/poses_bounds_cvd.npy            [F,17] LLFF poses + bounds
  raw/<scene>/dense/mv_images/<frame>/camXX.png     12 cameras per time step
  raw/<scene>/dense/mv_masks/<frame>/camXX.png      evaluation masks
  raw/<scene>/dense/images_<W>x288/<frame>.png      the monocular video (frame i = camera i % 12)
  masks/<scene>/dense/masks/final/<frame>_final.png dynamic masks (1-bit)
  depths/<scene>/disp/<frame>.npy                   disparity
  flows/<scene>/dense/flows/interval_k/<a>_<b>.npz  {flow, coord_diff}
"""
import pathlib

import numpy as np
import PIL.Image

SCENE = "Balloon1"
N_CAMS = 12
H, W, F = 288, 36, 14


def _pose(yaw_deg, pitch_deg, t):
    y, p = np.deg2rad(yaw_deg), np.deg2rad(pitch_deg)
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    c2w = np.eye(4)
    c2w[:3, :3] = Ry @ Rx
    c2w[:3, 3] = t
    return c2w


def opencv_c2w(i):
    """camera i (poses repeat with period 12 upstream; here every row is distinct)"""
    return _pose(2.5 * (i % N_CAMS) - 12, 0.7 * (i % 5), [0.11 * (i % N_CAMS) + 0.003 * i, 0.02 * (i % 3), 0.01 * i])


def build_tree(root, seed=20240607):
    root = pathlib.Path(root)
    rng = np.random.default_rng(seed)
    dense = root / "raw" / SCENE / "dense"
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    # LLFF rows: [down, right, back | t | (h, w, f)] per frame + near/far bounds
    rows = []
    for i in range(F):
        c2w = opencv_c2w(i)
        r, d, f = c2w[:3, 0], c2w[:3, 1], c2w[:3, 2]
        m = np.stack([d, r, -f, c2w[:3, 3], np.array([H * 2.0, W * 2.0, 0.9 * W * 2.0 + i])], 1)  # [3,5], double-res hwf
        rows.append(np.concatenate([m.reshape(-1), [0.5 + 0.01 * i, 9.0]]))
    dense.mkdir(parents=True)
    np.save(dense / "poses_bounds_cvd.npy", np.stack(rows))
    mono = dense / f"images_{W}x{H}"
    mono.mkdir()
    for f in range(F):
        (dense / "mv_images" / f"{f:05d}").mkdir(parents=True)
        (dense / "mv_masks" / f"{f:05d}").mkdir(parents=True)
        for c in range(N_CAMS):
            img = np.stack([127 + 100 * np.sin(xx / 5 + f + c), 127 + 100 * np.cos(yy / 17 + 0.3 * c), 30.0 * ((xx + yy + f) % 8)], -1)
            img = np.clip(img + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)
            PIL.Image.fromarray(img).save(dense / "mv_images" / f"{f:05d}" / f"cam{c + 1:02d}.png")
            if c == f % N_CAMS:
                PIL.Image.fromarray(img).save(mono / f"{f:05d}.png")
            em = (((xx - W * 0.5 - c) ** 2 + (yy - H * 0.4 - 3 * f) ** 2) < (0.3 * W) ** 2)
            PIL.Image.fromarray(np.repeat((em * 255).astype(np.uint8)[..., None], 3, -1)).save(
                dense / "mv_masks" / f"{f:05d}" / f"cam{c + 1:02d}.png")
    mdir = root / "masks" / SCENE / "dense" / "masks" / "final"
    ddir = root / "depths" / SCENE / "disp"
    mdir.mkdir(parents=True)
    ddir.mkdir(parents=True)
    for f in range(F):
        dyn = ((xx - W * (0.3 + 0.02 * f)) ** 2 + (yy - H * 0.5) ** 2) < (0.25 * W) ** 2
        PIL.Image.fromarray(dyn).save(mdir / f"{f:05d}_final.png")
        disp = (1.0 / (2.0 + 0.4 * np.sin(xx / W * 2 + f) + 0.2 * np.cos(yy / H * 5))).astype(np.float32)
        np.save(ddir / f"{f:05d}.npy", disp)
    for k in (1, 2):
        fdir = root / "flows" / SCENE / "dense" / "flows" / f"interval_{k}"
        fdir.mkdir(parents=True)
        for a in range(F):
            for b in (a - k, a + k):
                if 0 <= b < F:
                    flow = rng.normal(0, 1.5, (H, W, 2)).astype(np.float32)
                    cd = rng.normal(0, 0.6, (H, W, 2)).astype(np.float32)
                    np.savez(fdir / f"{a:05d}_{b:05d}.npz", flow=flow, coord_diff=cd)
    return root


# ---------------------------------------------------------------------------- monocular video layout
MONO_SCENE = "my_video"
MH, MW, MF = 40, 56, 10


def build_mono_tree(root, seed=424242):
    """The layout the reference's preprocessing writes for an in-the-wild monocular video and
    ``MonoVisualizationDataset`` reads (pgdvs/datasets/mono_vis.py:96-118,262-275,684-738):
      <scene>/rgbs/<name>.png, poses/<name>.npz {K[4,4], c2w[4,4]}, depths/<name>.npz {depth[H,W]},
      masks/final/<name>_final.png, flows/interval_k/<a>_<b>.npz {flow, coord_diff}"""
    root = pathlib.Path(root)
    rng = np.random.default_rng(seed)
    sd = root / MONO_SCENE
    for d in ("rgbs", "poses", "depths", "masks/final"):
        (sd / d).mkdir(parents=True)
    yy, xx = np.mgrid[0:MH, 0:MW].astype(np.float32)
    names = [f"{i:05d}" for i in range(MF)]
    for i, nm in enumerate(names):
        img = np.stack([127 + 100 * np.sin(xx / 6 + i), 127 + 100 * np.cos(yy / 9 + 0.4 * i), 25.0 * ((xx + 2 * yy + i) % 10)], -1)
        PIL.Image.fromarray(np.clip(img + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)).save(sd / "rgbs" / f"{nm}.png")
        K = np.eye(4)
        K[0, 0] = K[1, 1] = 0.9 * MW + 0.3 * i
        K[0, 2], K[1, 2] = MW / 2.0, MH / 2.0
        np.savez(sd / "poses" / f"{nm}.npz", K=K, c2w=_pose(3.0 * i - 12, 1.1 * i, [0.07 * i, 0.01 * i * i, 0.02 * i]))
        depth = (1.5 + 0.5 * np.sin(xx / MW * 3 + 0.5 * i) + 0.3 * np.cos(yy / MH * 4)).astype(np.float32)
        np.savez(sd / "depths" / f"{nm}.npz", depth=depth)
        dyn = ((xx - MW * (0.3 + 0.03 * i)) ** 2 + (yy - MH * 0.5) ** 2) < (0.2 * MW) ** 2
        PIL.Image.fromarray(dyn).save(sd / "masks" / "final" / f"{nm}_final.png")
    for k in (1, 2):
        fdir = sd / "flows" / f"interval_{k}"
        fdir.mkdir(parents=True)
        for a in range(MF):
            for b in (a - k, a + k):
                if 0 <= b < MF:
                    np.savez(fdir / f"{names[a]}_{names[b]}.npz", flow=rng.normal(0, 1.5, (MH, MW, 2)).astype(np.float32),
                             coord_diff=rng.normal(0, 0.6, (MH, MW, 2)).astype(np.float32))
    return root
