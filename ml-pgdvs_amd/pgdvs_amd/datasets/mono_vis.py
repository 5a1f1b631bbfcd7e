"""In-the-wild monocular video -> the renderer's ``data`` dict along a bullet-time camera path.

Mirror of ``pgdvs.datasets.mono_vis.MonoVisualizationDataset`` (pgdvs/datasets/mono_vis.py:34-738):
same constructor keywords, same per-scene directory layout (the one the reference's preprocessing
writes: ``rgbs/``, ``poses/*.npz {K, c2w}``, ``depths/*.npz {depth}``, ``masks/final/*_final.png``,
``flows/interval_k/<a>_<b>.npz {flow, coord_diff}``), same ``__getitem__`` keys and values.

The target cameras are not input cameras: ``n_render_frames`` time stamps are spread around
``vis_center_time``; at each, the pose is interpolated between the two neighbouring input poses
(linear translation, quaternion slerp exactly as the reference evaluates it -- without a
shortest-arc sign flip or re-normalisation, ``utils/geometry.py:468-515``) and composed with a
small circular "bullet-time" offset whose radius follows the scene's near depth
(``nvidia_vis.py:692-722``).

Host-side numpy / PIL input plumbing, like ``datasets/nvidia_eval.py``; resizes that upstream
does with OpenCV (only taken when files differ in size) use PIL filters.
"""
import pathlib
from math import acos, sin

import numpy as np
import PIL.Image
import torch
from torch.utils.data import Dataset

from .nvidia_eval import _resize, compute_pcl, depth_range_from_points, read_flow_npz

N_BT_REPS = 8


# ---------------------------------------------------------------------------- camera path
def rotmat_to_qvec(R):
    """(w, x, y, z) of a rotation matrix, COLMAP's eigenvector form (utils/geometry.py:448-465)."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R).flat
    K = np.array([
        [Rxx - Ryy - Rzz, 0, 0, 0],
        [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
        [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
        [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def qvec_to_rotmat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * y ** 2 - 2 * z ** 2, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x ** 2 - 2 * z ** 2, 2 * y * z - 2 * w * x],
        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x ** 2 - 2 * y ** 2]])


def interpolate_pose(c2w_a, c2w_b, ratio):
    """pose at ``ratio`` in [0,1] between two camera-to-world matrices (utils/geometry.py:468-515)"""
    qa, qb = rotmat_to_qvec(c2w_a[:3, :3]), rotmat_to_qvec(c2w_b[:3, :3])
    theta = acos(float(np.dot(qa, qb)))
    q = qa if theta == 0 else (sin((1 - ratio) * theta) * qa + sin(ratio * theta) * qb) / sin(theta)
    out = np.eye(4)
    out[:3, :3] = qvec_to_rotmat(q)
    out[:3, 3] = c2w_a[:3, 3] + (c2w_b[:3, 3] - c2w_a[:3, 3]) * ratio
    return out


def bullet_time_offsets(focal, num_frames, sc, max_disp):
    """inverse of small circular translations in the image plane (nvidia_vis.py:692-722)"""
    max_trans = (max_disp / sc if sc is not None else max_disp) / focal
    poses = []
    for i in range(num_frames):
        t = np.eye(4)
        t[0, 3] = max_trans * np.sin(2.0 * np.pi * float(i) / float(num_frames))
        t[1, 3] = max_trans * np.cos(2.0 * np.pi * float(i) / float(num_frames)) / 2.0
        poses.append(np.linalg.inv(t))
    return poses


def render_path(all_K, all_c2w, near_depths, *, vis_center_time, n_render_frames, vis_time_interval, vis_bt_max_disp):
    """[(time, index, c2w)] of the visualisation cameras of one scene (:113-197)"""
    n = all_K.shape[0]
    times = np.linspace(max(0, vis_center_time - vis_time_interval), min(n - 2, vis_center_time + vis_time_interval),
                        n_render_frames).tolist()
    bt_sc = 1.0 / (np.percentile(near_depths, 5) * 0.9)
    offsets = bullet_time_offsets(all_K[0, 0, 0], len(times) // N_BT_REPS, bt_sc, vis_bt_max_disp) * (N_BT_REPS + 1)
    path = []
    for i, t in enumerate(times):
        t0 = int(np.floor(t))
        path.append((t, i, interpolate_pose(all_c2w[t0], all_c2w[t0 + 1], t - np.floor(t)) @ offsets[i]))
    return path


def select_frames_for_time(tgt_time, n_frames, n_track_one_side):
    """the two input frames around a fractional time stamp and the tracker windows (:281-336; note
    the newer-side window starts AT the newer frame and is one longer, as upstream)"""
    older, newer = int(np.floor(tgt_time)), int(np.floor(tgt_time)) + 1
    temporal = sorted(set(([older] if tgt_time > 0 else []) + ([newer] if tgt_time < n_frames - 1 else [])))
    if len(temporal) == 1:
        temporal = temporal * 2
    n_actual = len(temporal)  # (counted after the placeholder, as upstream :300-305)
    fwd = [temporal[0]] * n_track_one_side
    fwd_actual = list(range(max(0, older - n_track_one_side), temporal[0])) if tgt_time > 0 else []
    fwd[: len(fwd_actual)] = fwd_actual
    bwd = [temporal[1]] * n_track_one_side
    bwd_actual = list(range(newer, min(n_frames, newer + 1 + n_track_one_side))) if tgt_time < n_frames - 1 else []
    bwd[: len(bwd_actual)] = bwd_actual
    return {"temporal": temporal, "n_actual_temporal": n_actual, "fwd2tgt": fwd, "n_actual_fwd2tgt": len(fwd_actual),
            "bwd2tgt": bwd, "n_actual_bwd2tgt": len(bwd_actual)}


# ---------------------------------------------------------------------------- dataset
class MonoVisualizationDataset(Dataset):
    dataset_name = "Monocular Visualization"
    dataset_fname = "mono_vis"

    def __init__(self, *, data_root, max_hw, mode, rgb_range="0_1", use_aug=False, scene_ids=None, n_src_views_spatial=10,
                 n_src_views_temporal_track_one_side=5, vis_center_time=50, n_render_frames=200, vis_time_interval=10,
                 vis_bt_max_disp=32, flow_consist_thres=1.0):
        assert max_hw == -1, f"We enforce to use raw resolution. However, we receive max_hw of {max_hw}"
        assert not use_aug and mode in ["vis"] and rgb_range == "0_1" and scene_ids is not None
        self.mode, self.max_hw, self.use_aug, self.rgb_range = mode, max_hw, use_aug, rgb_range
        self.n_src_views_spatial = n_src_views_spatial
        self.n_src_views_temporal_track_one_side = n_src_views_temporal_track_one_side
        self.flow_consist_thres = flow_consist_thres
        self.data_root = pathlib.Path(data_root)
        assert self.data_root.exists(), self.data_root
        self.c2w_dict, self.K_dict, self.valid_fs = {}, {}, []
        for scene in scene_ids:
            sd = self.data_root / scene
            cams = [np.load(f) for f in sorted((sd / "poses").glob("*.npz"))]
            all_K, all_c2w = np.array([c["K"] for c in cams]), np.array([c["c2w"] for c in cams])
            near = np.array([np.percentile(np.load(f)["depth"].reshape(-1), 5) for f in sorted((sd / "depths").glob("*.npz"))])
            self.c2w_dict[scene], self.K_dict[scene] = all_c2w.copy(), all_K.copy()
            for t, i, c2w in render_path(all_K, all_c2w, near, vis_center_time=vis_center_time, n_render_frames=n_render_frames,
                                         vis_time_interval=vis_time_interval, vis_bt_max_disp=vis_bt_max_disp):
                self.valid_fs.append([scene, sd, t, i, c2w, 1.0])

    def __len__(self):
        return len(self.valid_fs)

    def _source_view(self, scene_dir, img_f, c2w, K, tgt_shape):
        h, w = tgt_shape
        rgb = _resize(np.array(PIL.Image.open(img_f)), h, w, PIL.Image.Resampling.BOX).astype(np.float32) / 255.0
        name = pathlib.Path(img_f).stem
        mask = _resize(np.array(PIL.Image.open(scene_dir / "masks" / "final" / f"{name}_final.png")), h, w,
                       PIL.Image.Resampling.NEAREST).astype(np.float32)
        depth = _resize(np.load(scene_dir / "depths" / f"{name}.npz")["depth"], h, w, PIL.Image.Resampling.NEAREST)
        flat_cam = np.concatenate(([h, w], np.asarray(K).flatten(), np.asarray(c2w).flatten())).astype(np.float32)
        return {"rgb": rgb, "flat_cam": flat_cam, "dyn_mask": mask, "depth": depth, "dyn_rgb": rgb * mask[..., None],
                "static_rgb": rgb * (1 - mask[..., None]), "pcl": compute_pcl(h, w, K, c2w, depth)}

    def _stack_views(self, scene_dir, img_fs, frame_ids, all_c2w, all_K, tgt_shape):
        views = [self._source_view(scene_dir, img_fs[f], all_c2w[f], all_K[f], tgt_shape) for f in frame_ids]
        return {k: (np.concatenate if k == "pcl" else np.stack)([v[k] for v in views], axis=0) for k in views[0]}

    def _read_flow(self, scene_dir, img_fs, a, b, tgt_shape):
        if a == b:
            return np.zeros(list(tgt_shape) + [2], np.float32), np.zeros(tgt_shape, np.float32)
        flow, occ = read_flow_npz(scene_dir / "flows" / f"interval_{abs(b - a)}" / f"{img_fs[a].stem}_{img_fs[b].stem}.npz",
                                  self.flow_consist_thres)
        assert flow.shape[:2] == tuple(tgt_shape), (flow.shape, tgt_shape)
        return flow, occ

    def __getitem__(self, index):
        scene_id, scene_dir, tgt_time, tgt_idx, tgt_c2w, _ = self.valid_fs[index]
        exts = {ex for ex, f in PIL.Image.registered_extensions().items() if f in PIL.Image.OPEN}
        img_fs = sorted(f for f in (pathlib.Path(scene_dir) / "rgbs").iterdir() if f.suffix in exts)
        all_c2w, all_K = self.c2w_dict[scene_id].copy(), self.K_dict[scene_id].copy()
        n = len(img_fs)
        assert n == all_c2w.shape[0] == all_K.shape[0], (n, all_c2w.shape, all_K.shape)
        sel = select_frames_for_time(tgt_time, n, self.n_src_views_temporal_track_one_side)
        d = np.linalg.norm(tgt_c2w[None, :3, 3] - all_c2w[:, :3, 3], axis=1)
        spatial_ids = sorted(np.argsort(d)[: self.n_src_views_spatial].tolist())
        tgt_h, tgt_w = np.array(PIL.Image.open(img_fs[0])).shape[:2]
        shape = (tgt_h, tgt_w)
        F32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
        spatial = self._stack_views(scene_dir, img_fs, spatial_ids, all_c2w, all_K, shape)
        temporal = self._stack_views(scene_dir, img_fs, sel["temporal"], all_c2w, all_K, shape)
        flow_fwd, occ_fwd = self._read_flow(scene_dir, img_fs, sel["temporal"][0], sel["temporal"][1], shape)
        flow_bwd, occ_bwd = self._read_flow(scene_dir, img_fs, sel["temporal"][1], sel["temporal"][0], shape)
        item = {
            "scene_id": scene_id,
            "seq_ids": torch.LongTensor(np.array([tgt_time, *spatial_ids, *sel["temporal"]])),  # (time truncated, as upstream)
            "rgb_src_spatial": F32(spatial["rgb"]), "dyn_rgb_src_spatial": F32(spatial["dyn_rgb"]),
            "static_rgb_src_spatial": F32(spatial["static_rgb"]),
            "n_actual_temporal": torch.LongTensor([sel["n_actual_temporal"]]),
            "rgb_src_temporal": F32(temporal["rgb"]), "dyn_rgb_src_temporal": F32(temporal["dyn_rgb"]),
            "static_rgb_src_temporal": F32(temporal["static_rgb"]),
            "dyn_mask_src_spatial": F32(spatial["dyn_mask"])[..., None], "dyn_mask_src_temporal": F32(temporal["dyn_mask"])[..., None],
            "flow_fwd": F32(flow_fwd), "flow_fwd_occ_mask": F32(occ_fwd)[..., None],
            "flow_bwd": F32(flow_bwd), "flow_bwd_occ_mask": F32(occ_bwd)[..., None],
            "flat_cam_tgt": F32(np.concatenate(([tgt_h, tgt_w], all_K[0].flatten(), tgt_c2w.flatten()))),
            "flat_cam_src_spatial": F32(spatial["flat_cam"]), "flat_cam_src_temporal": F32(temporal["flat_cam"]),
            "depth_src_temporal": F32(temporal["depth"])[..., None],
            "depth_range": F32(depth_range_from_points(spatial["pcl"], tgt_c2w)),
            "time_tgt": torch.FloatTensor([tgt_time]), "time_src_temporal": torch.FloatTensor(sel["temporal"]),
            "misc": {"scene_id": scene_id, "tgt_time": tgt_time, "tgt_idx": tgt_idx},
        }
        for side, key in (("fwd2tgt", "n_actual_fwd2tgt"), ("bwd2tgt", "n_actual_bwd2tgt")):
            tr = self._stack_views(scene_dir, img_fs, sel[side], all_c2w, all_K, shape)
            sfx = f"src_temporal_track_{side}"
            item.update({
                f"n_actual_temporal_track_{side}": torch.LongTensor([sel[key]]),
                f"rgb_{sfx}": F32(tr["rgb"]), f"dyn_rgb_{sfx}": F32(tr["dyn_rgb"]), f"static_rgb_{sfx}": F32(tr["static_rgb"]),
                f"dyn_mask_{sfx}": F32(tr["dyn_mask"])[..., None], f"flat_cam_{sfx}": F32(tr["flat_cam"]),
                f"depth_{sfx}": F32(tr["depth"])[..., None], f"time_{sfx}": torch.FloatTensor(sel[side]),
            })
        return item
