#!/usr/bin/env python3
"""Golden vectors for row A17 (tracker-window point aggregation): the reference's
PGDVSDynamicTrackRenderer.compute_pcl_for_tgt and .prepare_data
(pgdvs/renderers/pgdvs_renderer_dyn_track.py:98-396, :599-764) run on synthetic tracks /
visibilities.  The point trackers themselves (TAPIR / CoTracker) are out of scope, so tracks are
inputs.  pytorch3d's knn_points is the exact brute-force stub of make_golden.py."""
import pathlib
import sys
import types

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def main():
    MG._install_stubs()
    # tracker networks (third-party, need jax/tree) are not imported
    from unittest.mock import MagicMock
    for m in ("pgdvs.models.tapnet", "pgdvs.models.tapnet.interface", "pgdvs.models.cotracker", "pgdvs.models.cotracker.interface"):
        sys.modules[m] = MagicMock()
    import pgdvs.renderers.pgdvs_renderer_dyn_track as RT

    T = torch.from_numpy
    rng = np.random.default_rng(4242)
    H, W = 24, 32
    n_fwd, n_close, n_bwd = 2, 2, 1
    N = n_fwd + n_close + n_bwd
    raw_times = np.array([3.0, 4.0, 5.0, 6.0, 7.0], np.float32)  # fwd.., closest.., bwd..
    time_tgt_raw = np.array([5.4], np.float32)
    cams = np.stack([MG._flat_cam(H, W, 0.9 * W * (1 + 0.01 * i), MG._pose(1.5 * i - 3, 0.4 * i, [0.03 * i, 0.01 * i, 0.0]))
                     for i in range(N)]).astype(np.float32)
    cam_tgt = MG._flat_cam(H, W, 0.9 * W, MG._pose(0.3, 0.2, [0.05, 0.01, 0.0])).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depths = np.stack([(2.0 + 0.4 * np.sin(xx / W * 3 + i) + 0.2 * np.cos(yy / H * 2)).astype(np.float32) for i in range(N)])[..., None]
    rgbs = rng.random((N, H, W, 3), dtype=np.float32)
    masks = (rng.random((N, H, W, 1)) < 0.2).astype(np.float32)

    # the batch dict PGDVSDynamicTrackRenderer.prepare_data consumes (B=1, padded to 3 per side)
    def pad(a, n):
        return np.concatenate([a, np.zeros((n - a.shape[0],) + a.shape[1:], a.dtype)], 0)[None]

    data = {
        "n_actual_temporal_track_fwd2tgt": np.array([[n_fwd]]), "n_actual_temporal": np.array([[n_close]]),
        "n_actual_temporal_track_bwd2tgt": np.array([[n_bwd]]), "time_tgt": time_tgt_raw[None]}
    for key, arr in (("rgb", rgbs), ("dyn_mask", masks), ("depth", depths), ("flat_cam", cams), ("time", raw_times)):
        data[f"{key}_src_temporal_track_fwd2tgt"] = pad(arr[:n_fwd], 3)
        data[f"{key}_src_temporal"] = pad(arr[n_fwd:n_fwd + n_close], 2)
        data[f"{key}_src_temporal_track_bwd2tgt"] = pad(arr[n_fwd + n_close:], 3)
    renderer = RT.PGDVSDynamicTrackRenderer.__new__(RT.PGDVSDynamicTrackRenderer)
    torch.nn.Module.__init__(renderer)
    dfk = renderer.prepare_data(0, {k: T(np.asarray(v)) for k, v in data.items()}, 3 * 2 + 2, "cpu")
    assert dfk["idx_temporal_closest"] == [2, 3] and dfk["idx_real_track"] == [0, 1, 4]

    out = {}
    for case, (n_pt, with_base, base_thres, knn) in enumerate([(400, True, 0.03, 4), (300, True, 0.012, 6), (200, False, None, 5), (350, True, 0.02, 6)]):
        # smooth tracks: a start position plus a slow drift, so that the interpolated cloud is a surface
        start = np.stack([rng.uniform(1.0, W - 6.0, n_pt), rng.uniform(1.0, H - 4.0, n_pt)], -1)
        drift = rng.normal(size=(n_pt, 1, 2)) * 0.15 + np.array([0.6, 0.3])
        tracks = (start[:, None] + drift * np.arange(N)[None, :, None] + rng.normal(size=(n_pt, N, 2)) * 0.2).astype(np.float32)
        tracks[-3:, :, 0] += W  # partly outside the frame: zero padding of both samplers
        vis = rng.random((n_pt, N)) < 0.7
        vis[:, 2:4] = rng.random((n_pt, 2)) < 0.15
        vis[:10, 2:4] = True          # visible in a closest frame -> not used
        vis[10:14] = False            # visible nowhere
        query = np.concatenate([rng.integers(0, N, (n_pt, 1)), tracks[:, 0, ::-1]], 1).astype(np.float32)
        base_n = 120
        base_pts = (rng.normal(size=(base_n, 3)) * np.array([0.6, 0.4, 0.3]) + np.array([0.0, 0.0, 2.2])).astype(np.float32)
        base_rgb = rng.random((base_n, 3), dtype=np.float32)
        base = {"pcl": T(base_pts) if with_base else None, "pcl_rgbs": T(base_rgb) if with_base else None,
                "pcl_nn_dist_thres": torch.tensor(base_thres) if base_thres is not None else None}
        rc = types.SimpleNamespace(dyn_pcl_outlier_knn=knn, dyn_pcl_track_track2base_thres_mult=50, dyn_pcl_outlier_std_thres=0.1)
        pcl, pcl_rgb = renderer.compute_pcl_for_tgt(
            data_for_track=dfk, query_pts=T(query), tracks=T(tracks), track_visibles=T(vis), render_cfg=rc,
            base_pcl_info=base, device="cpu")
        out.update({f"c{case}_tracks": tracks, f"c{case}_vis": vis, f"c{case}_query": query, f"c{case}_with_base": with_base,
                    f"c{case}_base_thres": np.float32(base_thres if base_thres is not None else np.nan), f"c{case}_knn": knn,
                    f"c{case}_base_pts": base_pts, f"c{case}_base_rgb": base_rgb, f"c{case}_out_pcl": pcl.numpy(),
                    f"c{case}_out_rgb": pcl_rgb.numpy()})
        print(case, "output cloud:", pcl.shape[0], "(base", base_n if with_base else 0, ")")
    np.savez_compressed(
        OUT / "track_pcl.npz", H=H, W=W, N=N, n_cases=4, cam_tgt=cam_tgt,
        **{"data_" + k: np.asarray(v) for k, v in data.items()},
        dfk_times=dfk["time_for_track"].numpy(), dfk_time_tgt=dfk["time_tgt"].numpy(),
        dfk_idx_closest=np.array(dfk["idx_temporal_closest"]), dfk_idx_real=np.array(dfk["idx_real_track"]),
        dfk_rgbs=dfk["rgbs_for_track"].numpy()[:N], dfk_depths=dfk["depths_for_track"].numpy(), dfk_cams=dfk["flat_cams_for_track"].numpy(),
        **out)
    print(f"  track_pcl.npz {(OUT / 'track_pcl.npz').stat().st_size / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
