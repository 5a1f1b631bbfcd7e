"""CPU ORACLE for the PGDVS per-target-view rendering hot path.

TEST INFRASTRUCTURE ONLY.  This module (and oracle/pgdvs_oracle.c behind it) is a
CPU restatement of the reference algorithm; it may be imported only by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg -- as the
checker / reported baseline, never as the product path.  The product
(``ml-pgdvs_amd/``) never imports it and fails loudly without its HIP library.

Parity pinning (see tests/golden/make_golden.py and tests/test_oracle_golden.py):
rows A1-A8, A11, A12 of SURVEY.md section 8a are pinned against outputs of the
reference itself (imported in the build container under ``sys.modules`` stubs).
Row A9 (pytorch3d 0.7.4 point rasteriser/compositor, an un-vendored third-party
dependency) is restated from its published algorithm: **parity unpinned** there.

All ``file:line`` citations are relative to the upstream apple/ml-pgdvs tree.
"""
from __future__ import annotations

import ctypes
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_BUILD = _HERE / "_build"
_LIB_PATH = _BUILD / "libpgdvs_oracle.so"
_SRC = _HERE / "pgdvs_oracle.c"

CAM_BLOCK = 80

_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_u8_p = ctypes.POINTER(ctypes.c_uint8)
_c_i32_p = ctypes.POINTER(ctypes.c_int32)
_c_i64_p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> pathlib.Path:
    """Compile the C restatement (gcc, no FMA contraction)."""
    if (
        not force
        and _LIB_PATH.exists()
        and _LIB_PATH.stat().st_mtime >= _SRC.stat().st_mtime
    ):
        return _LIB_PATH
    _BUILD.mkdir(parents=True, exist_ok=True)
    cmd = [
        "gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared",
        "-fPIC", "-o", str(_LIB_PATH), str(_SRC), "-lm",
    ]
    subprocess.run(cmd, check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
        _lib.orc_num_threads.restype = ctypes.c_int
        _lib.orc_cam_prep.restype = ctypes.c_int
        _lib.orc_inv_f64.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(lib().orc_num_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------
# camera helpers
# --------------------------------------------------------------------------
def cam_prep(flat_cam) -> np.ndarray:
    """flat_cam[34] -> derived constants block (orc_cam_prep)."""
    fc = _f32(flat_cam).reshape(34)
    blk = np.zeros(CAM_BLOCK, np.float32)
    rc = lib().orc_cam_prep(_p(fc, _c_float_p), _p(blk, _c_float_p))
    if rc != 0:
        raise ValueError(f"singular camera matrix (rc={rc})")
    return blk


def get_batched_rays(flat_cam, H, W, render_stride=1):
    """pgdvs_renderer_base.py:17-57 for batch_size=1."""
    blk = cam_prep(flat_cam)
    rh = (H + render_stride - 1) // render_stride
    rw = (W + render_stride - 1) // render_stride
    n = rh * rw
    ro = np.empty((n, 3), np.float32)
    rd = np.empty((n, 3), np.float32)
    uv = np.empty((n, 2), np.float32)
    lib().orc_get_rays(
        _p(blk, _c_float_p), H, W, render_stride, _p(ro, _c_float_p), _p(rd, _c_float_p),
        _p(uv, _c_float_p),
    )
    return ro, rd, uv, (rh, rw)


# --------------------------------------------------------------------------
# A4 statistical outlier filter (pgdvs_renderer_dyn.py:401-457,
# st_geo_renderer.py:32-69)
# --------------------------------------------------------------------------
def knn_mean_dist(pts, K) -> np.ndarray:
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty(pts.shape[0], np.float32)
    lib().orc_knn_mean_dist(
        _p(pts, _c_float_p), ctypes.c_int64(pts.shape[0]), int(K), _p(out, _c_float_p)
    )
    return out


def outlier_threshold(avg_nn_dist: np.ndarray, std_thres: float):
    """median (torch.median = lower median) + unbiased std * thres, fp32."""
    n = avg_nn_dist.shape[0]
    if n == 0:
        return np.float32(np.nan)
    med = np.partition(avg_nn_dist, (n - 1) // 2)[(n - 1) // 2]
    # torch.std: unbiased, accumulated in double on CPU then cast
    std = np.float32(np.std(avg_nn_dist.astype(np.float64), ddof=1)) if n > 1 else np.float32(np.nan)
    return np.float32(med + std * np.float32(std_thres))


# --------------------------------------------------------------------------
# A2/A3/A4/A5: compute_dyn_pcl (pgdvs_renderer_dyn.py:275-540), softsplat mode
# --------------------------------------------------------------------------
def compute_dyn_pcl(
    *, dyn_mask_1, rgb_1, depth_1, flow_12, flow_12_occ_mask, rgb_2, depth_2,
    flat_cam_1, flat_cam_2, flat_cam_tgt, time_1, time_2, time_tgt,
    dyn_render_use_flow_consistency=False, dyn_pcl_remove_outlier=False,
    dyn_pcl_outlier_knn=50, dyn_pcl_outlier_std_thres=0.1,
):
    """Returns dict(flow_1_to_tgt[H,W,2], valid_dyn_mask_1[H,W,1], pcl[n,3],
    pcl_rgbs[n,3], pcl_nn_dist_thres, avg_nn_dist[n_valid], flag_not_outlier)."""
    H, W = dyn_mask_1.shape[:2]
    P = H * W
    cam1, cam2, camt = cam_prep(flat_cam_1), cam_prep(flat_cam_2), cam_prep(flat_cam_tgt)
    m = _f32(dyn_mask_1).reshape(P)
    occ = _f32(flow_12_occ_mask).reshape(P)
    fl = _f32(flow_12).reshape(P, 2)
    d1, d2 = _f32(depth_1).reshape(P), _f32(depth_2).reshape(P)
    c1, c2 = _f32(rgb_1).reshape(P, 3), _f32(rgb_2).reshape(P, 3)
    mask_eff = np.zeros(P, np.uint8)
    valid = np.zeros(P, np.uint8)
    pcl = np.zeros((P, 3), np.float32)
    rgbf = np.zeros((P, 3), np.float32)
    lib().orc_dyn_warp(
        H, W, _p(m, _c_float_p), _p(occ, _c_float_p), int(bool(dyn_render_use_flow_consistency)),
        _p(fl, _c_float_p), _p(d1, _c_float_p), _p(d2, _c_float_p), _p(c1, _c_float_p),
        _p(c2, _c_float_p), _p(cam1, _c_float_p), _p(cam2, _c_float_p),
        ctypes.c_float(time_1), ctypes.c_float(time_2), ctypes.c_float(time_tgt),
        _p(mask_eff, _c_u8_p), _p(valid, _c_u8_p), _p(pcl, _c_float_p), _p(rgbf, _c_float_p),
    )
    vb = valid.astype(bool)
    dyn_pcl = pcl[vb]  # row-major compaction == boolean indexing order (:318-320)
    rgb_flow_12 = rgbf[vb]
    avg = knn_mean_dist(dyn_pcl, dyn_pcl_outlier_knn)
    thres = outlier_threshold(avg, dyn_pcl_outlier_std_thres)
    if dyn_pcl_remove_outlier:
        flag = avg < thres
    else:
        flag = np.ones(dyn_pcl.shape[0], bool)
    keep = np.zeros(P, np.uint8)
    keep[np.flatnonzero(vb)[flag]] = 1
    flow_1_to_tgt = np.zeros((P, 2), np.float32)
    valid_mask = np.zeros(P, np.float32)
    lib().orc_project_flow_dense(
        H, W, _p(camt, _c_float_p), _p(pcl, _c_float_p), _p(keep, _c_u8_p),
        _p(flow_1_to_tgt, _c_float_p), _p(valid_mask, _c_float_p),
    )
    return {
        "flow_1_to_tgt": flow_1_to_tgt.reshape(H, W, 2),
        "valid_dyn_mask_1": valid_mask.reshape(H, W, 1),
        "pcl": dyn_pcl[flag],
        "pcl_rgbs": rgb_flow_12[flag],
        "pcl_nn_dist_thres": thres,
        "avg_nn_dist": avg,
        "flag_not_outlier": flag,
        "valid": vb.reshape(H, W),
        "mask_eff": mask_eff.reshape(H, W),
        "pcl_dense": pcl.reshape(H, W, 3),
        "rgb_dense": rgbf.reshape(H, W, 3),
        "keep": keep.reshape(H, W),
    }


def project(flat_cam_tgt, pts):
    camt = cam_prep(flat_cam_tgt)
    pts = _f32(pts).reshape(-1, 3)
    uv = np.empty((pts.shape[0], 2), np.float32)
    lib().orc_project(_p(camt, _c_float_p), _p(pts, _c_float_p), ctypes.c_int64(pts.shape[0]), _p(uv, _c_float_p))
    return uv


# --------------------------------------------------------------------------
# A6 + A7: softsplat metric and forward splat
# --------------------------------------------------------------------------
def backwarp_l1(rgb1_chw, rgb2_chw, flow_chw) -> np.ndarray:
    """mean_c |rgb1 - backwarp(rgb2, flow)|  -> [H,W] (pgdvs_renderer_base.py:68-78,91-138)."""
    _, H, W = rgb1_chw.shape
    a, b, f = _f32(rgb1_chw), _f32(rgb2_chw), _f32(flow_chw)
    out = np.empty((H, W), np.float32)
    lib().orc_backwarp_l1(H, W, _p(a, _c_float_p), _p(b, _c_float_p), _p(f, _c_float_p), _p(out, _c_float_p))
    return out


def softsplat_raw(ten_in, ten_flow) -> np.ndarray:
    """softsplat_func.forward (softsplat.py:339-427): in[B,C,H,W], flow[B,2,H,W]."""
    ten_in, ten_flow = _f32(ten_in), _f32(ten_flow)
    B, C, H, W = ten_in.shape
    out = np.zeros_like(ten_in)
    lib().orc_softsplat_fwd(_p(ten_in, _c_float_p), _p(ten_flow, _c_float_p), _p(out, _c_float_p), B, C, H, W)
    return out


def softsplat_bwd_raw(ten_in, ten_flow, grad_out):
    """gradients of the raw splat wrt input and flow (softsplat.py:459-617)."""
    ten_in, ten_flow, grad_out = _f32(ten_in), _f32(ten_flow), _f32(grad_out)
    B, C, H, W = ten_in.shape
    gi = np.empty_like(ten_in)
    gf = np.empty_like(ten_flow)
    lib().orc_softsplat_bwd(_p(ten_in, _c_float_p), _p(ten_flow, _c_float_p), _p(grad_out, _c_float_p), _p(gi, _c_float_p),
                            _p(gf, _c_float_p), B, C, H, W)
    return gi, gf


def softsplat_corners(flow_2hw) -> np.ndarray:
    f = _f32(flow_2hw)
    _, H, W = f.shape
    idx = np.empty((H, W, 4), np.int32)
    lib().orc_softsplat_corners(_p(f, _c_float_p), H, W, _p(idx, _c_i32_p))
    return idx


def softsplat(ten_in, ten_flow, ten_metric, str_mode: str) -> np.ndarray:
    """softsplat.softsplat (softsplat.py:280-333)."""
    mode = str_mode.split("-")
    assert mode[0] in ["sum", "avg", "linear", "soft"]
    ten_in = _f32(ten_in)
    if mode[0] in ("sum", "avg"):
        assert ten_metric is None
    else:
        assert ten_metric is not None
        ten_metric = _f32(ten_metric)
    if str_mode == "avg":
        ten_in = np.concatenate([ten_in, np.ones_like(ten_in[:, :1])], 1)
    elif mode[0] == "linear":
        ten_in = np.concatenate([ten_in * ten_metric, ten_metric], 1)
    elif mode[0] == "soft":
        e = np.exp(ten_metric)
        ten_in = np.concatenate([ten_in * e, e], 1)
    out = softsplat_raw(ten_in, ten_flow)
    if mode[0] in ("avg", "linear", "soft"):
        norm = out[:, -1:, :, :]
        if len(mode) == 1 or mode[1] == "addeps":
            norm = norm + np.float32(0.0000001)
        elif mode[1] == "zeroeps":
            norm = norm.copy()
            norm[norm == 0.0] = 1.0
        elif mode[1] == "clipeps":
            norm = np.clip(norm, np.float32(0.0000001), None)
        out = out[:, :-1, :, :] / norm
    return out


def softsplat_img(rgb_src1, flow_src1_to_tgt, rgb_src2, flow_src1_to_src2, alpha, metric=None):
    """PGDVSBaseRenderer.softsplat_img (pgdvs_renderer_base.py:59-89); batch of NCHW."""
    if metric is None:
        metric = np.stack(
            [backwarp_l1(rgb_src1[b], rgb_src2[b], flow_src1_to_src2[b]) for b in range(rgb_src1.shape[0])]
        )[:, None]
    a = np.float32(alpha)
    m = np.clip(-a * metric, -a, a).astype(np.float32)
    return softsplat(rgb_src1, flow_src1_to_tgt, m, "soft"), metric


# --------------------------------------------------------------------------
# A8: PGDVSDynamicRenderer.forward, softsplat / pcl variants, no tracker
# (pgdvs_renderer_dyn.py:63-257)
# --------------------------------------------------------------------------
def dyn_forward(data: dict, render_cfg: dict, static_noise=None, alpha=100.0):
    """data holds numpy arrays with the reference's keys (batch dim included).
    static_noise[B,3,H,W] replaces torch.randn_like (:181) -- already un-clamped."""
    rgb_t = _f32(data["rgb_src_temporal"])
    B, _, H, W, _ = rgb_t.shape
    flow_1_to_tgt, dyn_mask_src_1, rgb_src_1, rgb_src_2, flow12 = [], [], [], [], []
    infos = []
    for b in range(B):
        m1 = _f32(data["dyn_mask_src_temporal"])[b, 0]
        if np.sum(m1) > 0:
            r = compute_dyn_pcl(
                dyn_mask_1=m1, rgb_1=rgb_t[b, 0], depth_1=data["depth_src_temporal"][b, 0],
                flow_12=data["flow_fwd"][b], flow_12_occ_mask=data["flow_fwd_occ_mask"][b],
                rgb_2=rgb_t[b, 1], depth_2=data["depth_src_temporal"][b, 1],
                flat_cam_1=data["flat_cam_src_temporal"][b, 0], flat_cam_2=data["flat_cam_src_temporal"][b, 1],
                flat_cam_tgt=data["flat_cam_tgt"][b],
                time_1=float(data["time_src_temporal"][b, 0]), time_2=float(data["time_src_temporal"][b, 1]),
                time_tgt=float(data["time_tgt"][b, 0]),
                dyn_render_use_flow_consistency=render_cfg["dyn_render_use_flow_consistency"],
                dyn_pcl_remove_outlier=render_cfg["dyn_pcl_remove_outlier"],
                dyn_pcl_outlier_knn=render_cfg["dyn_pcl_outlier_knn"],
                dyn_pcl_outlier_std_thres=render_cfg["dyn_pcl_outlier_std_thres"],
            )
            flow_1_to_tgt.append(r["flow_1_to_tgt"])
            rgb_src_1.append(rgb_t[b, 0])
            rgb_src_2.append(rgb_t[b, 1])
            dyn_mask_src_1.append(r["valid_dyn_mask_1"])
            infos.append(r)
        else:
            flow_1_to_tgt.append(np.zeros((H, W, 2), np.float32))
            rgb_src_1.append(np.zeros((H, W, 3), np.float32))
            rgb_src_2.append(np.zeros((H, W, 3), np.float32))
            dyn_mask_src_1.append(np.zeros((H, W, 1), np.float32))
            infos.append(None)
        flow12.append(_f32(data["flow_fwd"])[b])
    if render_cfg["dyn_render_type"] == "softsplat":
        mask = np.stack(dyn_mask_src_1).transpose(0, 3, 1, 2)
        f1t = np.stack(flow_1_to_tgt).transpose(0, 3, 1, 2)
        c2 = np.stack(rgb_src_2).transpose(0, 3, 1, 2)
        f12 = np.stack(flow12).transpose(0, 3, 1, 2)
        c1 = np.stack(rgb_src_1).transpose(0, 3, 1, 2)
        if static_noise is None:
            static_noise = np.zeros_like(c1)
        c1 = c1 * mask + np.clip(_f32(static_noise), 0.0, 1.0) * (1 - mask)
        c1 = _f32(c1)
        splat_full, metric = softsplat_img(c1, f1t, c2, f12, alpha)
        splat_mask, _ = softsplat_img(mask, f1t, c2, f12, alpha, metric=metric)
        render_dyn_mask = (splat_mask > 1e-3).astype(np.float32)
        render_dyn_rgb = splat_full * render_dyn_mask
        extra = {"splat_full": splat_full, "splat_mask": splat_mask, "metric": metric,
                 "flow_1_to_tgt": f1t, "valid_dyn_mask_1": mask, "rgb_src_1": c1}
    elif render_cfg["dyn_render_type"] == "pcl":
        rgbs, masks = [], []
        for b in range(B):
            if infos[b] is None:
                rgbs.append(np.zeros((H, W, 3), np.float32))
                masks.append(np.zeros((H, W, 1), np.float32))
            else:
                img, msk, _ = render_points(
                    infos[b]["pcl"], infos[b]["pcl_rgbs"], data["flat_cam_tgt"][b], H, W,
                    render_cfg["dyn_render_pcl_pt_radius"], render_cfg["dyn_render_pcl_pts_per_pixel"],
                )
                rgbs.append(img)
                masks.append(msk)
        render_dyn_rgb = np.stack(rgbs).transpose(0, 3, 1, 2)
        render_dyn_mask = np.stack(masks).transpose(0, 3, 1, 2)
        extra = {}
    elif render_cfg["dyn_render_type"] == "mesh":
        rgbs, masks, faces = [], [], []
        for b in range(B):
            if infos[b] is None:
                rgbs.append(np.zeros((H, W, 3), np.float32))
                masks.append(np.zeros((H, W, 1), np.float32))
                faces.append(None)
            else:
                img, msk, fc = mesh_render(infos[b]["keep"], infos[b]["pcl_dense"], infos[b]["rgb_dense"], data["flat_cam_tgt"][b])
                rgbs.append(img)
                masks.append(msk[..., None])
                faces.append(fc)
        render_dyn_rgb = np.stack(rgbs).transpose(0, 3, 1, 2)
        render_dyn_mask = np.stack(masks).transpose(0, 3, 1, 2)
        extra = {"mesh_faces": faces}
    else:
        raise NotImplementedError(render_cfg["dyn_render_type"])
    track_rgb = np.zeros_like(render_dyn_rgb)
    track_mask = np.zeros_like(render_dyn_mask)
    if render_cfg.get("dyn_render_track_temporal", "none") == "no_tgt":
        # render_with_track (pgdvs_renderer_dyn_track.py:27-96) with supplied tracks
        for b in range(B):
            tracks, vis = data["track_tracks"][b], data["track_visibles"][b]
            if tracks is None or tracks.shape[0] == 0:
                continue
            dft = track_prepare_data(data, b)
            base = infos[b]
            pcl, rgbs, tinfo = track_compute_pcl_for_tgt(
                dft, tracks, vis, render_cfg, None if base is None else base["pcl"],
                None if base is None else base["pcl_rgbs"], None if base is None else base["pcl_nn_dist_thres"])
            img, msk, _ = render_points(pcl, rgbs, data["flat_cam_tgt"][b], H, W, render_cfg["dyn_render_pcl_pt_radius"],
                                        render_cfg["dyn_render_pcl_pts_per_pixel"])
            track_rgb[b] = img.transpose(2, 0, 1)
            track_mask[b] = msk.transpose(2, 0, 1)
            extra.setdefault("track_infos", []).append(tinfo)
    mask_for_track = ((~(render_dyn_mask > 0)) & (track_mask > 0)).astype(np.float32)
    final_rgb = (1 - mask_for_track) * render_dyn_rgb + mask_for_track * track_rgb
    final_mask = ((render_dyn_mask > 0) | (track_mask > 0)).astype(np.float32)
    return final_rgb, final_mask, {"temporal_closest_rgb": render_dyn_rgb, "temporal_closest_mask": render_dyn_mask,
                                   "temporal_track_rgb": track_rgb, "temporal_track_mask": track_mask, "infos": infos, **extra}


# --------------------------------------------------------------------------
# A9: point rasteriser + compositor (pytorch3d semantics; st_geo_renderer.py:77-120)
# --------------------------------------------------------------------------
def rasterize_points(pts, flat_cam_tgt, H, W, radius, K):
    cam = cam_prep(flat_cam_tgt)
    pts = _f32(pts).reshape(-1, 3)
    N = pts.shape[0]
    ndc = np.empty((N, 3), np.float32)
    lib().orc_points_to_ndc(_p(cam, _c_float_p), H, W, _p(pts, _c_float_p), ctypes.c_int64(N), ctypes.c_int64(3), _p(ndc, _c_float_p))
    idx = np.empty((H, W, K), np.int64)
    zbuf = np.empty((H, W, K), np.float32)
    d2 = np.empty((H, W, K), np.float32)
    lib().orc_raster_points_naive(
        _p(ndc, _c_float_p), ctypes.c_int64(N), H, W, ctypes.c_float(radius), int(K),
        _p(idx, _c_i64_p), _p(zbuf, _c_float_p), _p(d2, _c_float_p),
    )
    return idx, zbuf, d2


def points_to_ndc(pts, flat_cam_tgt, H, W):
    cam = cam_prep(flat_cam_tgt)
    pts = _f32(pts).reshape(-1, 3)
    ndc = np.empty((pts.shape[0], 3), np.float32)
    lib().orc_points_to_ndc(_p(cam, _c_float_p), H, W, _p(pts, _c_float_p), ctypes.c_int64(pts.shape[0]), ctypes.c_int64(3),
                            _p(ndc, _c_float_p))
    return ndc


def rasterize_points_window(ndc, H, W, radius, K, y0, y1, x0, x1):
    """The naive rasteriser on the pixel window [y0,y1) x [x0,x1) of the H x W image, all points tested:
    (idx, zbuf, d2)[y1-y0, x1-x0, K] -- what the full-size call returns on that window."""
    ndc = _f32(ndc).reshape(-1, 3)
    sh = (y1 - y0, x1 - x0, K)
    idx, zbuf, d2 = np.empty(sh, np.int64), np.empty(sh, np.float32), np.empty(sh, np.float32)
    lib().orc_raster_points_window(_p(ndc, _c_float_p), ctypes.c_int64(ndc.shape[0]), H, W, ctypes.c_float(radius), int(K),
                                   int(y0), int(y1), int(x0), int(x1), _p(idx, _c_i64_p), _p(zbuf, _c_float_p), _p(d2, _c_float_p))
    return idx, zbuf, d2


def rasterize_points_pointmajor(ndc, H, W, radius, K):
    """The naive rasteriser's per-pixel lists over the FULL frame from a point-major sweep (same arithmetic,
    same insertion rule, points in index order): (idx, zbuf, d2)[H, W, K].  Seconds at 1080p x 3.5 M points,
    where the pixel-major loop needs hours; proven equal to it in tests/test_oracle_golden.py."""
    ndc = _f32(ndc).reshape(-1, 3)
    sh = (H, W, K)
    idx, zbuf, d2 = np.empty(sh, np.int64), np.empty(sh, np.float32), np.empty(sh, np.float32)
    lib().orc_raster_points_pointmajor(_p(ndc, _c_float_p), ctypes.c_int64(ndc.shape[0]), int(H), int(W), ctypes.c_float(radius),
                                       int(K), _p(idx, _c_i64_p), _p(zbuf, _c_float_p), _p(d2, _c_float_p))
    return idx, zbuf, d2


def composite(idx, d2, radius, feat):
    H, W, K = idx.shape
    out = np.empty((H, W, 3), np.float32)
    if feat is not None:
        feat = _f32(feat).reshape(-1, 3)
        fp = _p(feat, _c_float_p)
    else:
        fp = None
    lib().orc_norm_weighted_composite(
        _p(idx, _c_i64_p), _p(d2, _c_float_p), H, W, K, ctypes.c_float(radius), fp,
        ctypes.c_int64(3), 3, _p(out, _c_float_p),
    )
    return out


def render_points(pts, rgbs, flat_cam_tgt, H, W, radius, K):
    """rgb image, (ones-render > 0) mask, fragments (st_geo_renderer.py:81-120)."""
    pts = _f32(pts).reshape(-1, 3)
    if pts.shape[0] == 0:
        return np.zeros((H, W, 3), np.float32), np.zeros((H, W, 1), np.float32), None
    if float(H) * W * pts.shape[0] > 2e9:
        # the pixel-major loop would take minutes to hours: the point-major sweep returns the same lists
        # (tests/test_oracle_golden.py::test_pointmajor_raster_equals_naive)
        idx, zbuf, d2 = rasterize_points_pointmajor(points_to_ndc(pts, flat_cam_tgt, H, W), H, W, radius, K)
    else:
        idx, zbuf, d2 = rasterize_points(pts, flat_cam_tgt, H, W, radius, K)
    img = composite(idx, d2, radius, rgbs)
    ones = composite(idx, d2, radius, None)
    mask = (ones[..., :1] > 0.0).astype(np.float32)
    return img, mask, (idx, zbuf, d2)


def mesh_render(keep, pcl_dense, rgb_dense, flat_cam_tgt):
    """render_dyn_mesh (pgdvs_renderer_dyn.py:542-669): keep[H,W] = valid_dyn_mask_1, dense
    vertices / colours over the source frame -> img[H,W,3], mask[H,W], face[H,W] (int64)."""
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    H, W = keep.shape
    cam = cam_prep(flat_cam_tgt)
    pcl = _f32(pcl_dense).reshape(H * W, 3)
    rgb = _f32(rgb_dense).reshape(H * W, 3)
    img = np.empty((H, W, 3), np.float32)
    mask = np.empty((H, W), np.float32)
    face = np.empty((H, W), np.int64)
    lib().orc_mesh_render(_p(cam, _c_float_p), H, W, _p(keep, _c_u8_p), _p(pcl, _c_float_p), _p(rgb, _c_float_p),
                          _p(img, _c_float_p), _p(mask, _c_float_p), _p(face, _c_i64_p))
    return img, mask, face


def static_geo_forward(st_pcl_rgb, flat_cam_tgt, H, W, render_cfg):
    """StaticGeoPointRenderer.forward (st_geo_renderer.py:26-122)."""
    st = _f32(st_pcl_rgb)
    pcl, rgb = st[:, :3], st[:, 3:]
    if render_cfg["st_pcl_remove_outlier"]:
        avg = knn_mean_dist(pcl, render_cfg["st_pcl_outlier_knn"])
        thres = outlier_threshold(avg, render_cfg["st_pcl_outlier_std_thres"])
        flag = avg < thres
        pcl, rgb = pcl[flag], rgb[flag]
    return render_points(
        pcl, rgb, flat_cam_tgt, H, W, render_cfg["st_render_pcl_pt_radius"],
        render_cfg["st_render_pcl_pts_per_pixel"],
    )


# --------------------------------------------------------------------------
# A12: static cloud aggregation (nvidia_eval_pure_geo.py:183-277)
# --------------------------------------------------------------------------
def hwf_to_K(h, w, f) -> np.ndarray:
    """_hwf_to_K(normalized=False) (nvidia_eval.py:1013-1019), float64."""
    K = np.eye(3)
    K[0, 0] = f
    K[1, 1] = f
    K[0, 2] = w / 2.0
    K[1, 2] = h / 2.0
    return K


def compute_pcl(H, W, K3, c2w, depth) -> np.ndarray:
    """_compute_pcl (nvidia_eval.py:840-847): K, c2w -> fp32, integer pixel centres."""
    flat = np.zeros(34, np.float32)
    flat[0], flat[1] = H, W
    K4 = np.eye(4, dtype=np.float32)
    K4[:3, :3] = np.asarray(K3, np.float32)
    flat[2:18] = K4.reshape(-1)
    flat[18:34] = np.asarray(c2w, np.float32).reshape(-1)
    cam = cam_prep(flat)
    d = _f32(depth).reshape(-1)
    pcl = np.empty((H * W, 3), np.float32)
    lib().orc_compute_pcl(_p(cam, _c_float_p), H, W, _p(d, _c_float_p), _p(pcl, _c_float_p))
    return pcl


def inv_f64(a) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float64)
    n = a.shape[0]
    out = np.empty_like(a)
    rc = lib().orc_inv_f64(_p(a, _c_double_p), _p(out, _c_double_p), n)
    if rc != 0:
        raise ValueError("singular")
    return out


def static_proj_mask(pcl, K3, w2c, H, W) -> np.ndarray:
    """_compute_pcl_proj_mask (nvidia_eval_pure_geo.py:257-277) -> bool [H*W]."""
    pcl = _f32(pcl).reshape(-1, 3)
    K3 = np.ascontiguousarray(K3, np.float64)
    w2c = np.ascontiguousarray(w2c, np.float64)
    mask = np.zeros(H * W, np.uint8)
    lib().orc_static_proj_mask(
        _p(pcl, _c_float_p), ctypes.c_int64(pcl.shape[0]), ctypes.c_int64(3), _p(K3, _c_double_p),
        _p(w2c, _c_double_p), H, W, _p(mask, _c_u8_p),
    )
    return mask.astype(bool)


def aggregate_static_pcl(rgbs, depths, dyn_masks, K3s, c2ws) -> np.ndarray:
    """_aggregate_static_pcl (nvidia_eval_pure_geo.py:183-255) on in-memory frames.
    rgbs[S,H,W,3] in [0,1] fp32, depths[S,H,W] fp32, dyn_masks[S,H,W] bool,
    K3s[S,3,3] float64, c2ws[S,4,4] float64 -> st_pcl_rgb[Ns,6] fp32."""
    S, H, W = depths.shape
    st_pcl = np.zeros((0, 3), np.float32)
    st_rgb = np.zeros((0, 3), np.float32)
    for i in range(S):
        pcl = compute_pcl(H, W, K3s[i], c2ws[i], depths[i])
        st_mask = (~np.asarray(dyn_masks[i], bool)).reshape(-1)
        if i > 0:
            pm = static_proj_mask(st_pcl, K3s[i], inv_f64(np.asarray(c2ws[i], np.float64)), H, W)
            st_mask = st_mask & (~pm)
        st_pcl = np.concatenate((st_pcl, pcl[st_mask]), 0)
        st_rgb = np.concatenate((st_rgb, _f32(rgbs[i]).reshape(-1, 3)[st_mask]), 0)
    return np.concatenate((st_pcl, st_rgb), 1)


# --------------------------------------------------------------------------
# A11: final composite (pgdvs_renderer.py:169-178)
# --------------------------------------------------------------------------
def combine(static_rgb, render_dyn_rgb, render_dyn_mask):
    st = (1 - render_dyn_mask) * static_rgb
    dy = render_dyn_mask * render_dyn_rgb
    return (st + dy).astype(np.float32), st.astype(np.float32), dy.astype(np.float32)


def render_view(data: dict, render_cfg: dict, static_noise=None, alpha=100.0):
    """PGDVSRenderer.forward with the geometric static renderer or the rgb_gnt
    shortcut (pgdvs_renderer.py:83-180)."""
    B, _, H, W, _ = data["rgb_src_temporal"].shape
    ret = {}
    if "rgb_gnt" in data:
        static_rgb = _f32(data["rgb_gnt"]).transpose(0, 3, 1, 2)
        ret["static_coarse_rgb"] = static_rgb
    else:
        imgs, masks = [], []
        for b in range(B):
            img, msk, _ = static_geo_forward(data["st_pcl_rgb"][b], data["flat_cam_tgt"][b], H, W, render_cfg)
            imgs.append(img)
            masks.append(msk)
        static_rgb = np.stack(imgs).transpose(0, 3, 1, 2)
        ret["geo_static_rgb"] = static_rgb
        ret["geo_static_mask"] = np.stack(masks).transpose(0, 3, 1, 2)
    dyn_rgb, dyn_mask, info = dyn_forward(data, render_cfg, static_noise=static_noise, alpha=alpha)
    ret["render_dyn_rgb"] = dyn_rgb
    ret["render_dyn_mask"] = dyn_mask
    c, cs, cd = combine(static_rgb, dyn_rgb, dyn_mask)
    ret["combined_rgb"], ret["combined_rgb_static"], ret["combined_rgb_dyn"] = c, cs, cd
    ret["_info"] = info
    return ret


# --------------------------------------------------------------------------
# A17 tracker-window aggregation (pgdvs_renderer_dyn_track.py)
# --------------------------------------------------------------------------
def knn_cross_mean_dist(query, pts, KK) -> np.ndarray:
    """mean of the KK smallest squared distances query -> pts (all KK columns, :299-312)."""
    query = _f32(query).reshape(-1, 3)
    pts = _f32(pts).reshape(-1, 3)
    out = np.empty(query.shape[0], np.float32)
    lib().orc_knn_cross_mean_dist(
        _p(query, _c_float_p), ctypes.c_int64(query.shape[0]), _p(pts, _c_float_p),
        ctypes.c_int64(pts.shape[0]), int(KK), _p(out, _c_float_p))
    return out


def track_prepare_data(data: dict, i_b: int) -> dict:
    """prepare_data (:599-764) without the fixed-shape padding that only the tracker
    networks need: frames ordered [fwd2tgt..., temporally-closest..., bwd2tgt...]."""
    parts = {k: [] for k in ("rgb", "dyn_mask", "depth", "flat_cam", "time")}
    kind = []
    for suffix, nkey, kd in (("_track_fwd2tgt", "n_actual_temporal_track_fwd2tgt", 2), ("", "n_actual_temporal", 1),
                             ("_track_bwd2tgt", "n_actual_temporal_track_bwd2tgt", 2)):
        n = int(np.asarray(data[nkey])[i_b, 0])
        if n <= 0:
            continue
        for k in parts:
            parts[k].append(np.asarray(data[f"{k}_src_temporal{suffix}"])[i_b, :n])
        kind += [kd] * n
    times = np.concatenate(parts["time"]).astype(np.float32)
    t_min = times.min()
    return {
        "rgbs": _f32(np.concatenate(parts["rgb"])), "dyn_masks": _f32(np.concatenate(parts["dyn_mask"])),
        "depths": _f32(np.concatenate(parts["depth"])), "flat_cams": _f32(np.concatenate(parts["flat_cam"])),
        "times": times - t_min, "time_tgt": np.float32(np.asarray(data["time_tgt"], np.float32)[i_b, 0] - t_min),
        "kind": np.array(kind, np.uint8)}


def track_points(dft: dict, tracks, vis):
    """orc_track_points -> (valid[P] bool, pcl[P,3], rgb[P,3])."""
    tracks = _f32(tracks)
    P, N = tracks.shape[:2]
    vis = np.ascontiguousarray(vis, dtype=np.uint8)
    _, H, W, _ = dft["rgbs"].shape
    cams = np.stack([cam_prep(fc) for fc in dft["flat_cams"]])
    valid = np.zeros(P, np.uint8)
    pcl = np.zeros((P, 3), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    rgbs = _f32(dft["rgbs"])
    depths = _f32(dft["depths"]).reshape(N, H, W)
    kind = np.ascontiguousarray(dft["kind"], dtype=np.uint8)
    times = _f32(dft["times"])
    lib().orc_track_points(
        ctypes.c_int64(P), N, H, W, _p(tracks, _c_float_p), _p(vis, _c_u8_p), _p(kind, _c_u8_p), _p(times, _c_float_p),
        ctypes.c_float(float(dft["time_tgt"])), _p(rgbs, _c_float_p), _p(depths, _c_float_p), _p(cams, _c_float_p),
        _p(valid, _c_u8_p), _p(pcl, _c_float_p), _p(rgb, _c_float_p))
    return valid.astype(bool), pcl, rgb


def track_compute_pcl_for_tgt(dft: dict, tracks, vis, render_cfg: dict, base_pcl=None, base_rgb=None, base_thres=None):
    """compute_pcl_for_tgt (:98-396).  Returns (pcl, rgb, info)."""
    K = int(render_cfg["dyn_pcl_outlier_knn"])
    valid, pcl, rgb = track_points(dft, tracks, vis)
    info = {"valid": valid, "pcl_all": pcl, "rgb_all": rgb}
    pcl, rgb = pcl[valid], rgb[valid]
    if pcl.shape[0] == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), info
    if base_pcl is not None and base_pcl.shape[0] > 0:
        avg = knn_cross_mean_dist(pcl, base_pcl, K + 1)
        keep = avg < np.float32(np.float32(base_thres) * np.float32(render_cfg["dyn_pcl_track_track2base_thres_mult"]))
        info["avg_track2base"] = avg
        pcl, rgb = pcl[keep], rgb[keep]
    if pcl.shape[0] > 0:
        avg = knn_mean_dist(pcl, K)
        thres = np.float32(base_thres) if base_thres is not None else outlier_threshold(avg, render_cfg["dyn_pcl_outlier_std_thres"])
        info["avg_self"] = avg
        keep = avg < thres
        pcl, rgb = pcl[keep], rgb[keep]
    info["n_track"] = pcl.shape[0]
    if base_pcl is not None and pcl.shape[0] > 0:
        pcl = np.concatenate([pcl, _f32(base_pcl)], 0)
        rgb = np.concatenate([rgb, _f32(base_rgb)], 0)
    return pcl, rgb, info
