"""On-disk NVIDIA Dynamic Scenes sequence -> the renderer's ``data`` dict.

Mirror of ``pgdvs.datasets.nvidia_eval.NvidiaDynEvaluationDataset``
(pgdvs/datasets/nvidia_eval.py:59-1040) and of
``pgdvs.datasets.nvidia_eval_pure_geo.NvidiaDynPureGeoEvaluationDataset``
(nvidia_eval_pure_geo.py:41-470) for the evaluation configuration the benchmark scripts
use (raw resolution, no augmentation, DynIBaR disparities): same constructor keywords, same
directory layout, same ``__getitem__`` keys / shapes / value conventions, so a
``DataLoader`` over it feeds ``PGDVSRenderer.forward`` exactly like upstream's.

Host-side numpy + PIL only (this is input plumbing, not the hot path).  Differences, all
outside what the golden fixture exercises: no zip containers, no ZoeDepth variants
(``use_zoe_depth`` must be "none"), and resizes that upstream does with OpenCV (image
``INTER_AREA``, depth / evaluation mask ``INTER_NEAREST``; only taken when a file's size
differs from the 288-row target) use PIL's BOX / NEAREST filters.

The pure-geometry variant builds the static cloud once per scene with the HIP aggregator
(``aggregate_static_pcl``) instead of upstream's numpy loop.
"""
import pathlib
from collections import defaultdict

import numpy as np
import PIL.Image
import torch
from torch.utils.data import Dataset

from .static_aggregation import hwf_to_K

ALL_SCENE_IDS_NVIDIA_DYN = ["Balloon1", "Balloon2", "Jumping", "Playground", "Skating", "Truck", "Umbrella", "dynamicFace"]
N_CAMS = 12
TGT_HEIGHT = 288


# ---------------------------------------------------------------------------- file formats
def read_llff_cams(poses_bounds_path):
    """``poses_bounds_cvd.npy`` [F,17] -> (hwf [F,3] float32, c2w [F,4,4] float64, OpenCV axes)
    (:608-645).  Stored columns are [down, right, back | t | hwf]; LLFF's fix-up gives
    [right, up, back], the final sign flip [right, down, forward]."""
    arr = np.load(poses_bounds_path, allow_pickle=True)
    n = arr.shape[0]
    m = arr[:, :15].reshape(n, 3, 5)
    rot_t = np.concatenate([m[:, :, 1:2], -m[:, :, 0:1], m[:, :, 2:4]], axis=2).astype(np.float32)  # [F,3,4]
    hwf = m[:, :, 4].astype(np.float32)
    c2w = np.zeros((n, 4, 4), np.float64)
    c2w[:, :3, :] = rot_t
    c2w[:, 3, 3] = 1.0
    c2w[..., 1:3] *= -1.0
    return hwf, c2w


def read_flow_npz(path, occ_thres=1.0):
    """``flows/interval_k/<a>_<b>.npz`` {flow[H,W,2], coord_diff[H,W,2]} -> (flow, occlusion
    mask = sum|coord_diff| > thres as float32) (:957-1011)."""
    info = np.load(path)
    flow = info["flow"]
    occ = (np.sum(np.abs(info["coord_diff"]), axis=2) > occ_thres).astype(np.float32)
    return flow, occ


def select_temporal_frames(tgt_frame_id, tgt_cam_id, n_frames, n_track_one_side):
    """Temporally closest source frames and the tracker windows on either side (:250-318).
    Returns dict(temporal=[a,b], n_actual_temporal, fwd2tgt=[...], n_actual_fwd2tgt,
    bwd2tgt=[...], n_actual_bwd2tgt); the lists are padded with the nearest frame id."""
    in_mono = tgt_frame_id % N_CAMS == tgt_cam_id
    if in_mono:  # the target itself is a frame of the input video: its neighbours
        temporal = [f for f in (tgt_frame_id - 1, tgt_frame_id + 1) if 0 <= f < n_frames]
    else:  # another camera at the same instant: the input frame of that instant
        temporal = [tgt_frame_id]
    n_actual = len(temporal)
    if n_actual == 1:
        temporal = temporal * 2  # placeholder duplicate (:277-279)
    fwd = [temporal[0]] * n_track_one_side
    older = list(range(max(0, temporal[0] - n_track_one_side), temporal[0])) if tgt_frame_id > 0 else []
    fwd[: len(older)] = older
    bwd = [temporal[1]] * n_track_one_side
    newer = list(range(temporal[1] + 1, min(n_frames, temporal[1] + 1 + n_track_one_side))) if tgt_frame_id < n_frames - 1 else []
    bwd[: len(newer)] = newer
    return {"in_mono": in_mono, "temporal": temporal, "n_actual_temporal": n_actual, "fwd2tgt": fwd,
            "n_actual_fwd2tgt": len(older), "bwd2tgt": bwd, "n_actual_bwd2tgt": len(newer)}


def select_spatial_frames(tgt_frame_id, tgt_cam_id, n_frames, c2w_all, n_views):
    """The ``n_views`` input frames whose camera centres are nearest to the target camera, from
    the +-12-frame window around the target instant, returned in ascending frame order
    (:320-358)."""
    in_mono = tgt_frame_id % N_CAMS == tgt_cam_id
    lo, hi = max(0, tgt_frame_id - N_CAMS), min(n_frames, tgt_frame_id + N_CAMS)
    pool = [f for f in range(lo, hi) if not (in_mono and f == tgt_frame_id)]
    d = np.linalg.norm(c2w_all[tgt_cam_id, :3, 3][None, :] - c2w_all[pool, :3, 3], axis=1)
    order = np.argsort(d)
    return sorted(pool[i] for i in order[:n_views])


def compute_pcl(h, w, K, c2w, depth):
    """_compute_pcl (:840-847): fp32 rays through integer pixel centres times z-depth."""
    K32, c32 = np.asarray(K, np.float32), np.asarray(c2w, np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    pix = np.stack([u.reshape(-1), v.reshape(-1), np.ones(h * w, np.float32)], 0)
    M = c32[:3, :3] @ np.linalg.inv(K32[:3, :3]).astype(np.float32)
    rays_d = (M @ pix).T
    return c32[:3, 3][None, :] + rays_d * np.asarray(depth, np.float32).reshape(-1, 1)


def depth_range_from_points(pcl_world, c2w_tgt):
    """near = 0.8 * min z, far = 1.2 * 90th-percentile z of the spatial sources' points in the
    target camera (:446-456)."""
    homo = np.pad(pcl_world, ((0, 0), (0, 1)), "constant", constant_values=1)
    z = (np.linalg.inv(c2w_tgt) @ homo.T).T[:, 2]
    return np.array([max(1e-16, 0.8 * np.min(z)), max(2e-16, 1.2 * np.quantile(z, 0.9))])


def _resize(arr, h, w, resample):
    if arr.shape[0] == h and arr.shape[1] == w:
        return arr
    return np.array(PIL.Image.fromarray(arr).resize((w, h), resample=resample))


# ---------------------------------------------------------------------------- dataset
class NvidiaDynEvaluationDataset(Dataset):
    dataset_name = "NVIDIA_Dyn Eval"
    dataset_fname = "nvidia_eval"

    def __init__(self, *, data_root, raw_data_dir, depth_data_dir, mask_data_dir, flow_data_dir, max_hw, mode,
                 rgb_range="0_1", use_aug=False, scene_ids=None, n_src_views_spatial=10,
                 n_src_views_temporal_track_one_side=5, use_zoe_depth="none", zoe_depth_data_path=None,
                 flow_consist_thres=1.0):
        assert max_hw == -1, f"We enforce to use raw resolution. However, we receive max_hw of {max_hw}"
        assert not use_aug
        assert mode in ["eval"], mode
        assert rgb_range == "0_1", rgb_range
        if use_zoe_depth != "none":
            raise NotImplementedError("ZoeDepth inputs are not mirrored; use the DynIBaR disparities (use_zoe_depth='none')")
        self.mode, self.max_hw, self.use_aug, self.rgb_range = mode, max_hw, use_aug, rgb_range
        self.n_src_views_spatial = n_src_views_spatial
        self.n_src_views_temporal_track_one_side = n_src_views_temporal_track_one_side
        self.flow_consist_thres = flow_consist_thres
        root = pathlib.Path(data_root)
        self.raw_data_dir, self.depth_data_dir = root / raw_data_dir, root / depth_data_dir
        self.mask_data_dir, self.flow_data_dir = root / mask_data_dir, root / flow_data_dir
        for d in (self.raw_data_dir, self.depth_data_dir, self.mask_data_dir, self.flow_data_dir):
            assert d.exists(), d
        scene_ids = ALL_SCENE_IDS_NVIDIA_DYN if scene_ids is None else scene_ids
        exts = {ex for ex, f in PIL.Image.registered_extensions().items() if f in PIL.Image.OPEN}
        # e.g. Balloon1/dense/mv_images/00000/cam01.jpg
        self.scene_img_dict = defaultdict(lambda: defaultdict(dict))
        entries = set()
        for f in self.raw_data_dir.glob("*/dense/mv_images/*/*"):
            if f.suffix not in exts:
                continue
            scene = f.parents[3].name
            if scene not in scene_ids:
                continue
            frame_id, cam_id = int(f.parent.name), int(f.stem.split("cam")[1]) - 1  # cameras are 1-based on disk
            self.scene_img_dict[scene][frame_id][cam_id] = str(f)
            entries.add((scene, str(self.raw_data_dir / scene / "dense"), frame_id, cam_id, str(f)))
        self.scene_img_dict = {k: dict(v) for k, v in self.scene_img_dict.items()}
        self.valid_fs = sorted(entries)  # same order on every worker / rank
        self._cam_cache = {}

    def __len__(self):
        return len(self.valid_fs)

    # ------------------------------------------------------------------ readers
    def _read_cam(self, scene_id):
        if scene_id not in self._cam_cache:
            hwf, c2w = read_llff_cams(self.raw_data_dir / scene_id / "dense" / "poses_bounds_cvd.npy")
            assert len(self.scene_img_dict[scene_id]) == hwf.shape[0], (len(self.scene_img_dict[scene_id]), hwf.shape[0])
            self._cam_cache[scene_id] = (hwf, c2w)
        hwf, c2w = self._cam_cache[scene_id]
        return hwf.copy(), c2w.copy()

    def _read_mask(self, scene_id, frame_id, tgt_h, tgt_w):
        m = np.array(PIL.Image.open(self.mask_data_dir / scene_id / "dense" / "masks" / "final" / f"{frame_id:05d}_final.png"))
        return _resize(m, tgt_h, tgt_w, PIL.Image.Resampling.NEAREST)  # True = dynamic

    def _read_depth(self, scene_id, frame_id):
        return 1 / (np.load(self.depth_data_dir / scene_id / "disp" / f"{frame_id:05d}.npy") + 1e-8)

    def _read_flow(self, scene_id, src_frame_id, tgt_frame_id, tgt_shape):
        if src_frame_id == tgt_frame_id:
            return np.zeros(list(tgt_shape) + [2], np.float32), np.zeros(tgt_shape, np.float32)
        k = abs(tgt_frame_id - src_frame_id)
        flow, occ = read_flow_npz(self.flow_data_dir / scene_id / "dense" / "flows" / f"interval_{k}" /
                                  f"{src_frame_id:05d}_{tgt_frame_id:05d}.npz", self.flow_consist_thres)
        assert flow.shape[:2] == tuple(tgt_shape), (flow.shape, tgt_shape)
        return flow, occ

    def _read_eval_mask(self, scene_id, frame_id, cam_id, h, w):
        f = self.raw_data_dir / scene_id / "dense" / "mv_masks" / f"{frame_id:05d}" / f"cam{cam_id + 1:02d}.png"
        m = np.float32(np.array(PIL.Image.open(f).convert("RGB"))[..., ::-1] > 1e-3)  # channel order as cv2.imread
        return _resize(m, h, w, PIL.Image.Resampling.NEAREST)

    def _target_rgb(self, scene_dir, img_f):
        """the target image; multi-view frames stored at another height are brought to the mono
        video's size with LANCZOS as upstream (:367-380)"""
        raw = np.array(PIL.Image.open(img_f))
        if raw.shape[0] != TGT_HEIGHT:
            mono = list(pathlib.Path(scene_dir).glob(f"images_*x{TGT_HEIGHT}"))
            assert len(mono) == 1, mono
            new_w, new_h = (int(x) for x in mono[0].name.split("images_")[1].split("x"))
            raw = np.array(PIL.Image.fromarray(raw).resize((new_w, new_h), resample=PIL.Image.Resampling.LANCZOS))
        assert raw.shape[0] == TGT_HEIGHT, raw.shape
        return raw

    # ------------------------------------------------------------------ one source view
    def _source_view(self, scene_id, frame_id, c2w, hwf, tgt_shape, with_geometry=True, img_f=None):
        """image, flat camera and (optionally) dynamic mask / depth / world points of an input
        frame (:728-838).  Frame i of the monocular video is camera i % 12 of time step i."""
        h, w = tgt_shape
        if img_f is None:
            img_f = self.scene_img_dict[scene_id][frame_id][frame_id % N_CAMS]
        rgb = _resize(np.array(PIL.Image.open(img_f)), h, w, PIL.Image.Resampling.BOX).astype(np.float32) / 255.0
        K = np.eye(4)
        K[:3, :3] = hwf_to_K(*hwf, tgt_shape=tgt_shape)
        flat_cam = np.concatenate(([h, w], K.flatten(), np.asarray(c2w).flatten())).astype(np.float32)
        out = {"rgb": rgb, "flat_cam": flat_cam}
        if with_geometry:
            mask = self._read_mask(scene_id, frame_id, h, w).astype(np.float32)
            depth = _resize(self._read_depth(scene_id, frame_id), h, w, PIL.Image.Resampling.NEAREST)
            out.update(dyn_mask=mask, depth=depth, dyn_rgb=rgb * mask[..., None], static_rgb=rgb * (1 - mask[..., None]),
                       pcl=compute_pcl(h, w, K, c2w, depth))
        return out

    def _stack_views(self, scene_id, frame_ids, all_c2w, all_hwf, tgt_shape):
        views = [self._source_view(scene_id, f, all_c2w[f], all_hwf[f], tgt_shape) for f in frame_ids]
        return {k: (np.concatenate if k == "pcl" else np.stack)([v[k] for v in views], axis=0) for k in views[0]}

    # ------------------------------------------------------------------ item
    def _common_item(self, index):
        scene_id, scene_dir, tgt_frame_id, tgt_cam_id, img_f = self.valid_fs[index]
        all_hwf, all_c2w = self._read_cam(scene_id)
        n_frames = all_hwf.shape[0]
        sel = select_temporal_frames(tgt_frame_id, tgt_cam_id, n_frames, getattr(self, "n_src_views_temporal_track_one_side", 0))
        raw_rgb = self._target_rgb(scene_dir, img_f)
        raw_h, raw_w = raw_rgb.shape[:2]
        all_hwf[:, 0], all_hwf[:, 1] = raw_h, raw_w  # the stored h, w belong to the full-resolution capture (:399-401)
        tgt_shape = (raw_h, raw_w)
        # NOTE upstream indexes the poses with the CAMERA id: the 12 camera poses repeat (:319-323)
        tgt = self._source_view(scene_id, tgt_frame_id, all_c2w[tgt_cam_id], all_hwf[tgt_cam_id], tgt_shape,
                                with_geometry=False, img_f=img_f)
        temporal = self._stack_views(scene_id, sel["temporal"], all_c2w, all_hwf, tgt_shape)
        flow_fwd, occ_fwd = self._read_flow(scene_id, sel["temporal"][0], sel["temporal"][1], tgt_shape)
        flow_bwd, occ_bwd = self._read_flow(scene_id, sel["temporal"][1], sel["temporal"][0], tgt_shape)
        T = torch.from_numpy
        F32 = lambda a: T(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
        item = {
            "scene_id": scene_id,
            "rgb_tgt": F32(tgt["rgb"]),
            "n_actual_temporal": torch.LongTensor([sel["n_actual_temporal"]]),
            "rgb_src_temporal": F32(temporal["rgb"]), "dyn_rgb_src_temporal": F32(temporal["dyn_rgb"]),
            "static_rgb_src_temporal": F32(temporal["static_rgb"]),
            "dyn_mask_src_temporal": F32(temporal["dyn_mask"])[..., None],
            "eval_mask": F32(self._read_eval_mask(scene_id, tgt_frame_id, tgt_cam_id, raw_h, raw_w)),
            "flow_fwd": F32(flow_fwd), "flow_fwd_occ_mask": F32(occ_fwd)[..., None],
            "flow_bwd": F32(flow_bwd), "flow_bwd_occ_mask": F32(occ_bwd)[..., None],
            "flat_cam_tgt": F32(tgt["flat_cam"]), "flat_cam_src_temporal": F32(temporal["flat_cam"]),
            "depth_src_temporal": F32(temporal["depth"])[..., None],
            "time_tgt": torch.FloatTensor([tgt_frame_id]), "time_src_temporal": torch.FloatTensor(sel["temporal"]),
            "misc": {"scene_id": scene_id, "tgt_frame_id": tgt_frame_id, "tgt_cam_id": tgt_cam_id},
        }
        ctx = dict(scene_id=scene_id, tgt_frame_id=tgt_frame_id, tgt_cam_id=tgt_cam_id, all_hwf=all_hwf, all_c2w=all_c2w,
                   n_frames=n_frames, sel=sel, tgt_shape=tgt_shape, F32=F32)
        return item, ctx

    def __getitem__(self, index):
        item, c = self._common_item(index)
        F32, sel, scene_id = c["F32"], c["sel"], c["scene_id"]
        spatial_ids = select_spatial_frames(c["tgt_frame_id"], c["tgt_cam_id"], c["n_frames"], c["all_c2w"], self.n_src_views_spatial)
        assert self.n_src_views_spatial < N_CAMS * 2
        spatial = self._stack_views(scene_id, spatial_ids, c["all_c2w"], c["all_hwf"], c["tgt_shape"])
        item["seq_ids"] = torch.LongTensor(np.array([c["tgt_frame_id"], *spatial_ids, *sel["temporal"]]))
        item.update({
            "rgb_src_spatial": F32(spatial["rgb"]), "dyn_rgb_src_spatial": F32(spatial["dyn_rgb"]),
            "static_rgb_src_spatial": F32(spatial["static_rgb"]), "dyn_mask_src_spatial": F32(spatial["dyn_mask"])[..., None],
            "flat_cam_src_spatial": F32(spatial["flat_cam"]), "depth_src_spatial": F32(spatial["depth"])[..., None],
            "depth_range": F32(depth_range_from_points(spatial["pcl"], c["all_c2w"][c["tgt_cam_id"]])),
        })
        for side, key in (("fwd2tgt", "n_actual_fwd2tgt"), ("bwd2tgt", "n_actual_bwd2tgt")):
            tr = self._stack_views(scene_id, sel[side], c["all_c2w"], c["all_hwf"], c["tgt_shape"])
            sfx = f"src_temporal_track_{side}"
            item.update({
                f"n_actual_temporal_track_{side}": torch.LongTensor([sel[key]]),
                f"rgb_{sfx}": F32(tr["rgb"]), f"dyn_rgb_{sfx}": F32(tr["dyn_rgb"]), f"static_rgb_{sfx}": F32(tr["static_rgb"]),
                f"dyn_mask_{sfx}": F32(tr["dyn_mask"])[..., None], f"flat_cam_{sfx}": F32(tr["flat_cam"]),
                f"depth_{sfx}": F32(tr["depth"])[..., None], f"time_{sfx}": torch.FloatTensor(sel[side]),
            })
        return item


class NvidiaDynPureGeoEvaluationDataset(NvidiaDynEvaluationDataset):
    """nvidia_eval_pure_geo.py:41-470 -- no spatial sources / tracker windows, plus the static
    cloud ``st_pcl_rgb`` of the whole monocular video, aggregated once per scene on the GPU
    (``device``; upstream: numpy at construction time, :166-178)."""
    dataset_name = "NVIDIA_Dyn Pure Geometry Eval"
    dataset_fname = "nvidia_eval_pure_geo"

    def __init__(self, *, data_root, raw_data_dir, depth_data_dir, mask_data_dir, flow_data_dir, max_hw, mode,
                 rgb_range="0_1", use_aug=False, scene_ids=None, flow_consist_thres=1.0, device="cuda"):
        super().__init__(data_root=data_root, raw_data_dir=raw_data_dir, depth_data_dir=depth_data_dir,
                         mask_data_dir=mask_data_dir, flow_data_dir=flow_data_dir, max_hw=max_hw, mode=mode,
                         rgb_range=rgb_range, use_aug=use_aug, scene_ids=scene_ids, n_src_views_spatial=0,
                         n_src_views_temporal_track_one_side=0, flow_consist_thres=flow_consist_thres)
        self.device = device
        self.st_pcl_dict = {scene: self._aggregate_static_pcl(scene) for scene in sorted(self.scene_img_dict)}

    def _load_mono_video(self, scene_id):
        """frames, depths, dynamic masks and cameras of the monocular video (:183-222)"""
        scene_dir = self.raw_data_dir / scene_id / "dense"
        mono = list(scene_dir.glob(f"images_*x{TGT_HEIGHT}"))
        assert len(mono) == 1, mono
        tgt_w, tgt_h = (int(x) for x in mono[0].name.split("images_")[1].split("x"))
        all_hwf, all_c2w = self._read_cam(scene_id)
        all_hwf[:, 0], all_hwf[:, 1] = tgt_h, tgt_w
        n = all_hwf.shape[0]
        imgs = np.stack([_resize(np.array(PIL.Image.open(mono[0] / f"{i:05d}.png")), tgt_h, tgt_w, PIL.Image.Resampling.LANCZOS)
                         for i in range(n)]).astype(np.float32) / 255.0
        depths = np.stack([self._read_depth(scene_id, i) for i in range(n)]).astype(np.float32)
        masks = np.stack([self._read_mask(scene_id, i, tgt_h, tgt_w).astype(bool) for i in range(n)])
        K3s = np.stack([hwf_to_K(*all_hwf[i]) for i in range(n)])
        return imgs, depths, masks, K3s, all_c2w

    def _aggregate_static_pcl(self, scene_id):
        from .static_aggregation import aggregate_static_pcl

        imgs, depths, masks, K3s, c2ws = self._load_mono_video(scene_id)
        dev = self.device
        cloud = aggregate_static_pcl(torch.from_numpy(imgs).to(dev), torch.from_numpy(depths).to(dev),
                                     torch.from_numpy(masks).to(dev), K3s, c2ws)
        return cloud.cpu()

    def __getitem__(self, index):
        item, c = self._common_item(index)
        item["seq_ids"] = torch.LongTensor(np.array([c["tgt_frame_id"], *c["sel"]["temporal"]]))
        item["st_pcl_rgb"] = self.st_pcl_dict[c["scene_id"]]  # [#pt, 6]: xyz, rgb
        return item
