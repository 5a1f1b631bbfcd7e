#!/usr/bin/env python3
"""Golden vectors for the on-disk formats -> ``data`` dict row: the reference's
own dataset classes (pgdvs/datasets/nvidia_eval.py NvidiaDynEvaluationDataset,
nvidia_eval_pure_geo.py NvidiaDynPureGeoEvaluationDataset) pointed at the synthetic tree of
nvidia_tree.py.  cv2 is not installed here: a two-function stand-in (imread / resize for the
equal-size case the tree exercises) is injected; everything else runs unmodified."""
import pathlib
import sys
import tempfile
import types

import numpy as np
import PIL.Image

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402
import nvidia_tree as NT  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
ITEMS = [(5, 5), (0, 0), (13, 1), (6, 2), (0, 3), (13, 7)]  # (frame, camera); camera == frame % 12: inside the mono video


MONO_KW = dict(n_src_views_spatial=3, n_src_views_temporal_track_one_side=2, vis_center_time=4, n_render_frames=16,
               vis_time_interval=3, vis_bt_max_disp=8, flow_consist_thres=1.0)
MONO_ITEMS = [0, 5, 9, 15]


def _cv2_stub():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_AREA = 0, 3

    def imread(path):
        return np.array(PIL.Image.open(path).convert("RGB"))[..., ::-1].copy()  # BGR like OpenCV

    def resize(img, dsize, interpolation=None):
        w, h = dsize
        assert img.shape[0] == h and img.shape[1] == w, "the fixture only exercises equal-size resizes"
        return img

    cv2.imread, cv2.resize = imread, resize
    return cv2


def _to_np(v):
    return v.numpy() if hasattr(v, "numpy") else v


def digest(a):
    """order-sensitive fingerprint of a bulky array (the loaders pass file contents through, so
    the fixture keeps fingerprints instead of megabytes): [dot with fixed weights, sum, min, max]"""
    a = np.asarray(a, np.float64).reshape(-1)
    w = np.random.default_rng(12345).random(a.size)
    return np.array([a @ w, a.sum(), a.min(), a.max()])


def main():
    MG._install_stubs()
    sys.modules["cv2"] = _cv2_stub()
    import pgdvs.datasets.nvidia_eval as NE
    import pgdvs.datasets.nvidia_eval_pure_geo as PG

    out = {}
    with tempfile.TemporaryDirectory() as td:
        NT.build_tree(td)
        kw = dict(data_root=td, raw_data_dir="raw", depth_data_dir="depths", mask_data_dir="masks", flow_data_dir="flows",
                  max_hw=-1, mode="eval", scene_ids=[NT.SCENE])
        ds = NE.NvidiaDynEvaluationDataset(n_src_views_spatial=4, n_src_views_temporal_track_one_side=2, flow_consist_thres=1.0, **kw)
        assert len(ds) == NT.F * NT.N_CAMS
        hwf, c2w = ds._read_cam(NT.SCENE)
        out["cam_hwf"], out["cam_c2w"] = hwf, c2w
        for n, (f, c) in enumerate(ITEMS):
            item = ds[f * NT.N_CAMS + c]
            assert item["misc"]["tgt_frame_id"] == f and item["misc"]["tgt_cam_id"] == c
            for k, v in item.items():
                if k in ("scene_id", "misc"):
                    continue
                v = _to_np(v)
                if k.startswith("dyn_rgb") or k.startswith("static_rgb"):
                    continue  # = rgb * mask / rgb * (1 - mask): checked from those in the test
                if k.startswith("rgb_"):
                    q = np.round(v * 255.0)
                    assert np.abs(q / 255.0 - v).max() < 1e-6
                    v = q.astype(np.uint8)  # exact: the loader divides uint8 by 255
                elif "mask" in k:
                    assert set(np.unique(v)) <= {0.0, 1.0}
                    v = v.astype(np.uint8)
                if v.size > 2048:
                    out[f"i{n}_{k}__shape"] = np.array(v.shape)
                    out[f"i{n}_{k}__digest"] = digest(v)
                else:
                    out[f"i{n}_{k}"] = v
        pg = PG.NvidiaDynPureGeoEvaluationDataset(flow_consist_thres=1.0, **kw)
        item = pg[5 * NT.N_CAMS + 5]
        out["pg_keys"] = np.array(sorted(k for k in item.keys()))
        st = _to_np(item["st_pcl_rgb"])
        out["pg_st_pcl_rgb__shape"], out["pg_st_pcl_rgb__digest"], out["pg_st_pcl_rgb_head"] = np.array(st.shape), digest(st), st[:64]
        out["pg_flat_cam_tgt"] = _to_np(item["flat_cam_tgt"])
    np.savez_compressed(OUT / "nvidia_items.npz", items=np.array(ITEMS), **out)
    print(f"  nvidia_items.npz {(OUT / 'nvidia_items.npz').stat().st_size / 1024:.1f} KiB, keys {len(out)}")

    # ---- monocular-video visualisation dataset (bullet-time camera path between the input poses)
    if not hasattr(np, "mat"):  # the reference's quaternion helper predates NumPy 2 (geometry.py:123)
        np.mat = np.asmatrix
    import pgdvs.datasets.mono_vis as MV

    out = {}
    with tempfile.TemporaryDirectory() as td:
        NT.build_mono_tree(td)
        ds = MV.MonoVisualizationDataset(data_root=td, max_hw=-1, mode="vis", scene_ids=[NT.MONO_SCENE], **MONO_KW)
        out["n_items"] = len(ds)
        out["all_tgt_c2w"] = np.stack([e[4] for e in ds.valid_fs])
        out["all_tgt_time"] = np.array([e[2] for e in ds.valid_fs])
        for n, idx in enumerate(MONO_ITEMS):
            item = ds[idx]
            for k, v in item.items():
                if k in ("scene_id", "misc") or k.startswith("dyn_rgb") or k.startswith("static_rgb"):
                    continue
                v = _to_np(v)
                if k.startswith("rgb_"):
                    v = np.round(v * 255.0).astype(np.uint8)
                elif "mask" in k:
                    v = v.astype(np.uint8)
                if v.size > 2048:
                    out[f"i{n}_{k}__shape"], out[f"i{n}_{k}__digest"] = np.array(v.shape), digest(v)
                else:
                    out[f"i{n}_{k}"] = v
    np.savez_compressed(OUT / "mono_items.npz", items=np.array(MONO_ITEMS), **out)
    print(f"  mono_items.npz {(OUT / 'mono_items.npz').stat().st_size / 1024:.1f} KiB, keys {len(out)}")


if __name__ == "__main__":
    main()
