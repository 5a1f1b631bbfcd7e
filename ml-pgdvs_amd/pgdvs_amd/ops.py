"""torch-tensor front end of the C ABI (include/pgdvs_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
function below only validates tensors, allocates outputs / scratch with
``torch.empty`` and enqueues hand-written HIP kernels through ctypes.  Nothing in
this module computes on the CPU; CPU tensors are rejected.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CAM_BLOCK, PgdvsHipError, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us, 13
    # times per view); the raw handle of the current device's current stream is one C call
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise PgdvsHipError(
            f"{name}: tensor is on {t.device}; the MI355X path has no CPU fallback "
            "(move the data dict to the GPU as pgdvs.engines does)"
        )
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _rows(t: torch.Tensor, name: str) -> torch.Tensor:
    """2-D fp32 GPU tensor whose rows are contiguous (row stride arbitrary): views such as
    ``cloud[:, 3:]`` are passed to the kernels as (pointer, row stride) without a copy."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise PgdvsHipError(f"{name}: expected a GPU tensor (no CPU fallback)")
    assert t.ndim == 2 and t.shape[1] >= 3, (name, t.shape)
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[0] > 0 and t.stride(1) != 1:
        t = t.contiguous()
    return t


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _ws(nbytes: int, device) -> torch.Tensor:
    if int(nbytes) < 0:  # the *_workspace_bytes entry points return a negative status for shapes they reject
        msg = _lib.load().pgdvs_last_error().decode("utf-8", "replace")
        raise PgdvsHipError(f"workspace query rejected the shape (status {int(nbytes)}): {msg}")
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------
def cam_prep(flat_cams: torch.Tensor) -> torch.Tensor:
    """flat_cams[..., 34] -> camera blocks[..., 80]."""
    fc = _req(flat_cams, torch.float32, "flat_cams")
    assert fc.shape[-1] == 34, fc.shape
    n = fc.numel() // 34
    out = torch.empty(fc.shape[:-1] + (CAM_BLOCK,), dtype=torch.float32, device=fc.device)
    check(_lib.load().pgdvs_cam_prep(_ptr(fc), n, _ptr(out), _stream()), "pgdvs_cam_prep")
    return out


def get_rays(cam_block: torch.Tensor, H: int, W: int, stride: int = 1):
    cam = _req(cam_block, torch.float32, "cam_block")
    rh, rw = (H + stride - 1) // stride, (W + stride - 1) // stride
    n = rh * rw
    ro = torch.empty((n, 3), dtype=torch.float32, device=cam.device)
    rd = torch.empty((n, 3), dtype=torch.float32, device=cam.device)
    uv = torch.empty((n, 2), dtype=torch.float32, device=cam.device)
    check(_lib.load().pgdvs_get_rays(_ptr(cam), H, W, stride, _ptr(ro), _ptr(rd), _ptr(uv), _stream()), "pgdvs_get_rays")
    return ro, rd, uv, (rh, rw)


def dyn_warp(dyn_mask1, occ, use_fc, flow12, depth1, depth2, rgb1, rgb2, cam1, cam2, times):
    H, W = dyn_mask1.shape[0], dyn_mask1.shape[1]
    dev = dyn_mask1.device
    m = _req(dyn_mask1, torch.float32, "dyn_mask1")
    o = _req(occ, torch.float32, "occ") if occ is not None else None
    args = [_req(t, torch.float32, n) for t, n in [
        (flow12, "flow12"), (depth1, "depth1"), (depth2, "depth2"), (rgb1, "rgb1"), (rgb2, "rgb2"),
        (cam1, "cam1"), (cam2, "cam2"), (times, "times")]]
    mask_eff = torch.empty((H, W), dtype=torch.uint8, device=dev)
    valid = torch.empty((H, W), dtype=torch.uint8, device=dev)
    pcl = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    rgbf = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    check(_lib.load().pgdvs_dyn_warp(
        H, W, _ptr(m), _ptr(o), int(bool(use_fc)), *[_ptr(a) for a in args],
        _ptr(mask_eff), _ptr(valid), _ptr(pcl), _ptr(rgbf), _stream()), "pgdvs_dyn_warp")
    return mask_eff, valid, pcl, rgbf


def compact_u8(flags: torch.Tensor):
    """-> (idx[int32, capacity n], count[int32, 1]) both on the device."""
    f = _req(flags, torch.uint8, "flags").reshape(-1)
    n = f.numel()
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=f.device)
    cnt = torch.empty(1, dtype=torch.int32, device=f.device)
    if n == 0:
        cnt.zero_()
        return idx, cnt
    lib = _lib.load()
    ws = _ws(lib.pgdvs_compact_workspace_bytes(n), f.device)
    check(lib.pgdvs_compact_u8(_ptr(f), n, _ptr(idx), _ptr(cnt), _ptr(ws), ws.numel(), _stream()), "pgdvs_compact_u8")
    return idx, cnt


def gather_rows(src: torch.Tensor, idx: torch.Tensor, cnt: torch.Tensor) -> torch.Tensor:
    s = _req(src, torch.float32, "src")
    width = s.shape[-1]
    cap = idx.numel()
    dst = torch.empty((cap, width), dtype=torch.float32, device=s.device)
    check(_lib.load().pgdvs_gather_rows(_ptr(s), _ptr(idx), _ptr(cnt), cap, width, _ptr(dst), _stream()), "pgdvs_gather_rows")
    return dst


def knn_mean_dist(pts: torch.Tensor, cnt: torch.Tensor, K: int, algo: int = 0) -> torch.Tensor:
    """algo: 0 auto (grid when K+1 <= 64), 1 brute force, 2 grid."""
    p = _req(pts, torch.float32, "pts").reshape(-1, 3)
    c = _req(cnt, torch.int32, "count")
    out = torch.empty(max(p.shape[0], 1), dtype=torch.float32, device=p.device)
    lib = _lib.load()
    ws = _ws(lib.pgdvs_knn_workspace_bytes(p.shape[0]), p.device)
    check(lib.pgdvs_knn_mean_dist(_ptr(p), _ptr(c), p.shape[0], int(K), _ptr(out), int(algo), _ptr(ws), ws.numel(),
                                  _stream()), "pgdvs_knn_mean_dist")
    return out


def outlier_flags(avg: torch.Tensor, cnt: torch.Tensor, std_thres: float, remove_outlier: bool):
    a = _req(avg, torch.float32, "avg")
    thres = torch.empty(1, dtype=torch.float32, device=a.device)
    flag = torch.empty(max(a.numel(), 1), dtype=torch.uint8, device=a.device)
    lib = _lib.load()
    ws = _ws(lib.pgdvs_outlier_workspace_bytes(a.numel()), a.device)
    check(lib.pgdvs_outlier_flags(_ptr(a), _ptr(cnt), a.numel(), float(std_thres), int(bool(remove_outlier)),
                                  _ptr(thres), _ptr(flag), _ptr(ws), ws.numel(), _stream()), "pgdvs_outlier_flags")
    return thres, flag


def knn_cross_mean_dist(queries: torch.Tensor, qcnt: torch.Tensor, pts: torch.Tensor, cnt: torch.Tensor, KK: int) -> torch.Tensor:
    """mean of the KK smallest squared distances from each query to ``pts`` (all columns)."""
    q = _req(queries, torch.float32, "queries").reshape(-1, 3)
    p = _req(pts, torch.float32, "pts").reshape(-1, 3)
    out = torch.empty(max(q.shape[0], 1), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    ws = _ws(lib.pgdvs_knn_cross_workspace_bytes(p.shape[0], q.shape[0]), q.device)
    check(lib.pgdvs_knn_cross_mean_dist(_ptr(q), _ptr(_req(qcnt, torch.int32, "query_count")), q.shape[0], _ptr(p),
                                        _ptr(_req(cnt, torch.int32, "count")), p.shape[0], int(KK), _ptr(out), _ptr(ws),
                                        ws.numel(), _stream()), "pgdvs_knn_cross_mean_dist")
    return out


def threshold_flags(avg, cnt, thres, mult: float = 1.0, alt_thres=None, gate_count=None) -> torch.Tensor:
    a = _req(avg, torch.float32, "avg")
    flag = torch.empty(max(a.numel(), 1), dtype=torch.uint8, device=a.device)
    check(_lib.load().pgdvs_threshold_flags(_ptr(a), _ptr(cnt), a.numel(), _ptr(_req(thres, torch.float32, "thres")), float(mult),
                                            _ptr(alt_thres), _ptr(gate_count), _ptr(flag), _stream()), "pgdvs_threshold_flags")
    return flag


def concat_rows(a, cnt_a, b=None, cnt_b=None, require_a: bool = False):
    """[a[:cnt_a], b[:cnt_b]] with device-side counts -> (rows, count)."""
    a = _req(a, torch.float32, "a")
    assert a.ndim == 2
    nb = 0
    if b is not None:
        b = _req(b, torch.float32, "b")
        assert b.ndim == 2 and b.shape[1] == a.shape[1], (a.shape, b.shape)
        nb = b.shape[0]
    out = torch.empty((max(a.shape[0] + nb, 1), a.shape[1]), dtype=torch.float32, device=a.device)
    cnt = torch.empty(1, dtype=torch.int32, device=a.device)
    check(_lib.load().pgdvs_concat_rows(_ptr(a), _ptr(cnt_a), a.shape[0], _ptr(b), _ptr(cnt_b), nb, a.shape[1],
                                        int(bool(require_a)), _ptr(out), _ptr(cnt), _stream()), "pgdvs_concat_rows")
    return out, cnt


def track_points(tracks, visibles, frame_kind, times, time_tgt, rgbs, depths, cams):
    """A17 per-track validity, 3-D point at the target time and colour.
    tracks[P,N,2], visibles[P,N] bool, frame_kind: N host ints (1 closest / 2 track frame),
    times[N], time_tgt[1] (raw, device), rgbs[N,H,W,3], depths[N,H,W], cams[N,80]."""
    t = _req(tracks, torch.float32, "tracks")
    P, N = t.shape[0], t.shape[1]
    v = visibles
    if not v.is_cuda:
        raise PgdvsHipError("visibles: expected a GPU tensor (no CPU fallback)")
    v = v.contiguous().view(torch.uint8) if v.dtype == torch.bool else _req(v, torch.uint8, "visibles")
    rg = _req(rgbs, torch.float32, "rgbs")
    H, W = rg.shape[1], rg.shape[2]
    kind = (C.c_uint8 * N)(*[int(k) for k in frame_kind])
    valid = torch.empty(max(P, 1), dtype=torch.uint8, device=t.device)
    pcl = torch.empty((max(P, 1), 3), dtype=torch.float32, device=t.device)
    rgb = torch.empty((max(P, 1), 3), dtype=torch.float32, device=t.device)
    check(_lib.load().pgdvs_track_points(_ptr(t), _ptr(v), P, N, C.cast(kind, C.c_void_p), _ptr(_req(times, torch.float32, "times")),
                                         _ptr(_req(time_tgt, torch.float32, "time_tgt")), _ptr(rg),
                                         _ptr(_req(depths, torch.float32, "depths")), H, W, _ptr(_req(cams, torch.float32, "cams")),
                                         _ptr(valid), _ptr(pcl), _ptr(rgb), _stream()), "pgdvs_track_points")
    return valid[:P], pcl[:P], rgb[:P]


def scatter_keep(idx, flag, cnt, P: int) -> torch.Tensor:
    keep = torch.empty(P, dtype=torch.uint8, device=idx.device)
    check(_lib.load().pgdvs_scatter_keep(_ptr(idx), _ptr(flag), _ptr(cnt), idx.numel(), _ptr(keep), P, _stream()), "pgdvs_scatter_keep")
    return keep


def project_flow_dense(cam_tgt, pcl, keep, H: int, W: int):
    cam = _req(cam_tgt, torch.float32, "cam_tgt")
    flow = torch.empty((2, H, W), dtype=torch.float32, device=cam.device)
    mask = torch.empty((H, W), dtype=torch.float32, device=cam.device)
    check(_lib.load().pgdvs_project_flow_dense(H, W, _ptr(cam), _ptr(_req(pcl, torch.float32, "pcl")),
                                               _ptr(_req(keep, torch.uint8, "keep")), _ptr(flow), _ptr(mask), _stream()),
          "pgdvs_project_flow_dense")
    return flow, mask


def project_points(cam_tgt, pts):
    p = _req(pts, torch.float32, "pts").reshape(-1, 3)
    uv = torch.empty((p.shape[0], 2), dtype=torch.float32, device=p.device)
    check(_lib.load().pgdvs_project_points(_ptr(_req(cam_tgt, torch.float32, "cam_tgt")), _ptr(p), p.shape[0], _ptr(uv), _stream()),
          "pgdvs_project_points")
    return uv


def backwarp_l1(rgb1, rgb2, flow):
    a, b, f = _req(rgb1, torch.float32, "rgb1"), _req(rgb2, torch.float32, "rgb2"), _req(flow, torch.float32, "flow")
    B, _, H, W = a.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=a.device)
    check(_lib.load().pgdvs_backwarp_l1(_ptr(a), _ptr(b), _ptr(f), _ptr(out), B, H, W, _stream()), "pgdvs_backwarp_l1")
    return out


_MODES = {"sum": 0, "avg": 1, "linear": 2, "soft": 3}
_EPS = {"addeps": 0, "zeroeps": 1, "clipeps": 2}


def softsplat_fwd(ten_in, ten_flow, ten_metric, mode: int, eps: int):
    x = _req(ten_in, torch.float32, "tenIn")
    f = _req(ten_flow, torch.float32, "tenFlow")
    m = _req(ten_metric, torch.float32, "tenMetric") if ten_metric is not None else None
    B, Cc, H, W = x.shape
    assert f.shape == (B, 2, H, W), f.shape
    out = torch.empty_like(x)
    lib = _lib.load()
    ws = _ws(lib.pgdvs_softsplat_workspace_bytes(B, Cc, H, W, mode), x.device)
    check(lib.pgdvs_softsplat_fwd(_ptr(x), _ptr(f), _ptr(m), _ptr(out), B, Cc, H, W, mode, eps, _ptr(ws), ws.numel(), _stream()),
          "pgdvs_softsplat_fwd")
    return out


def softsplat_bwd(ten_in, ten_flow, grad_out, need_in: bool = True, need_flow: bool = True):
    """gradients of the raw ("sum") splat wrt input and flow."""
    i = _req(ten_in, torch.float32, "tenIn")
    f = _req(ten_flow, torch.float32, "tenFlow")
    g = _req(grad_out, torch.float32, "tenOutgrad")
    B, Cc, H, W = i.shape
    gi = torch.empty_like(i) if need_in else None
    gf = torch.empty_like(f) if need_flow else None
    check(_lib.load().pgdvs_softsplat_bwd(_ptr(i), _ptr(f), _ptr(g), _ptr(gi), _ptr(gf), B, Cc, H, W, _stream()),
          "pgdvs_softsplat_bwd")
    return gi, gf


def splat_rng_state(device, seed: int | None = None) -> torch.Tensor:
    """Device-resident state {seed, draw number} of the splat kernel's own noise (``dyn_splat_composite(rng_state=...)``);
    the seed defaults to torch's (``torch.initial_seed()``)."""
    s = torch.initial_seed() if seed is None else int(seed)
    return torch.tensor([s & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)


def splat_noise_field(rng_state, H: int, W: int) -> torch.Tensor:
    """[3,H,W]: the un-clamped normal field the NEXT ``dyn_splat_composite(rng_state=rng_state)`` call draws."""
    st = _req(rng_state, torch.int64, "rng_state")
    out = torch.empty((3, H, W), dtype=torch.float32, device=st.device)
    check(_lib.load().pgdvs_splat_noise_field(H, W, _ptr(st), _ptr(out), _stream()), "pgdvs_splat_noise_field")
    return out


def dyn_splat_composite(rgb1, rgb2, flow12, flow_1_to_tgt, valid_mask, noise, alpha, static_rgb, out_combined=None,
                        rng_state=None):
    """Returns planar (render_dyn_rgb[3,H,W], render_dyn_mask[H,W], combined, combined_static, combined_dyn).
    ``out_combined``: optional caller-owned contiguous [3,H,W] fp32 buffer the kernel writes the composite into
    (e.g. a slice of the caller's image stack: no copy afterwards).  ``noise`` [3,H,W]: the injected normal field
    (parity tests); ``noise=None`` with ``rng_state`` (``splat_rng_state``): the kernel draws it itself and advances
    the state; both None: zeros."""
    r1 = _req(rgb1, torch.float32, "rgb1")
    H, W = r1.shape[0], r1.shape[1]
    dev = r1.device
    dyn_rgb = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    dyn_mask = torch.empty((H, W), dtype=torch.float32, device=dev)
    st = _req(static_rgb, torch.float32, "static_rgb") if static_rgb is not None else None
    comb = None
    if st is not None:
        if out_combined is not None:
            if not (out_combined.is_cuda and out_combined.dtype == torch.float32 and out_combined.is_contiguous()
                    and tuple(out_combined.shape) == (3, H, W)):
                raise PgdvsHipError(f"out_combined: expected a contiguous fp32 GPU tensor [3,{H},{W}], got {tuple(out_combined.shape)}")
            two = torch.empty((2, 3, H, W), dtype=torch.float32, device=dev)
            comb = (out_combined, two[0], two[1])
        else:
            three = torch.empty((3, 3, H, W), dtype=torch.float32, device=dev)
            comb = (three[0], three[1], three[2])
    nz = _req(noise, torch.float32, "noise") if noise is not None else None
    lib = _lib.load()
    ws = _ws(lib.pgdvs_dyn_splat_workspace_bytes(H, W), dev)
    draw = nz is None and rng_state is not None
    fn = lib.pgdvs_dyn_splat_composite_rng if draw else lib.pgdvs_dyn_splat_composite
    check(fn(
        H, W, _ptr(r1), _ptr(_req(rgb2, torch.float32, "rgb2")), _ptr(_req(flow12, torch.float32, "flow12")),
        _ptr(_req(flow_1_to_tgt, torch.float32, "flow_1_to_tgt")), _ptr(_req(valid_mask, torch.float32, "valid_mask")),
        _ptr(_req(rng_state, torch.int64, "rng_state") if draw else nz), float(alpha), _ptr(st), _ptr(dyn_rgb), _ptr(dyn_mask),
        _ptr(comb[0] if comb is not None else None), _ptr(comb[1] if comb is not None else None),
        _ptr(comb[2] if comb is not None else None), _ptr(ws), ws.numel(), _stream()),
        "pgdvs_dyn_splat_composite_rng" if draw else "pgdvs_dyn_splat_composite")
    if comb is None:
        return dyn_rgb, dyn_mask, None, None, None
    return dyn_rgb, dyn_mask, comb[0], comb[1], comb[2]


def points_raster(pts, feat, cam_tgt, radius: float, K: int, H: int, W: int, *, n_points_dev=None,
                  want_fragments: bool = False, rgb_planar: bool = False, want_rgb: bool = True, row_bound: int | None = None):
    """pts[N,>=3] (xyz in the first 3 columns of each row), feat[N,>=3] rows.
    Returns dict(rgb, mask[, idx, zbuf, dist2]).  ``row_bound``: size the workspace for that many rows instead of N
    (capacity-sized cloud buffers with a device count: ``n_points_dev``); the dict then carries ``status`` (device
    int32[1]: 0 fine, 1 = the count exceeded the bound and rows were cut off, 2 = negative count; ``check_raster_status``)."""
    p = _rows(pts, "pts")
    n = p.shape[0]
    dev = p.device
    ft = _rows(feat, "feat") if feat is not None else None
    cam = _req(cam_tgt, torch.float32, "cam_tgt")
    idx = torch.empty((H, W, K), dtype=torch.int64, device=dev) if want_fragments else None
    zbuf = torch.empty((H, W, K), dtype=torch.float32, device=dev) if want_fragments else None
    d2 = torch.empty((H, W, K), dtype=torch.float32, device=dev) if want_fragments else None
    rgb = None
    if want_rgb and ft is not None:
        rgb = torch.empty((3, H, W) if rgb_planar else (H, W, 3), dtype=torch.float32, device=dev)
    mask = torch.empty((H, W), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if row_bound is None:
        ws = _ws(lib.pgdvs_points_raster_workspace_bytes(n, H, W, float(radius)), dev)
        check(lib.pgdvs_points_raster(
            _ptr(p), p.stride(0) if n else 3, _ptr(ft), ft.stride(0) if (ft is not None and n) else 3, n,
            _ptr(n_points_dev), _ptr(cam),
            float(radius), int(K), H, W, _ptr(idx), _ptr(zbuf), _ptr(d2), _ptr(rgb), int(bool(rgb_planar)), _ptr(mask),
            _ptr(ws), ws.numel(), _stream()), "pgdvs_points_raster")
        return {"rgb": rgb, "mask": mask, "idx": idx, "zbuf": zbuf, "dist2": d2}
    bound = max(0, min(int(row_bound), n))
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _ws(lib.pgdvs_points_raster_workspace_bytes(bound, H, W, float(radius)), dev)
    check(lib.pgdvs_points_raster_bounded(
        _ptr(p), p.stride(0) if n else 3, _ptr(ft), ft.stride(0) if (ft is not None and n) else 3, n,
        _ptr(n_points_dev), bound, _ptr(status), _ptr(cam),
        float(radius), int(K), H, W, _ptr(idx), _ptr(zbuf), _ptr(d2), _ptr(rgb), int(bool(rgb_planar)), _ptr(mask),
        _ptr(ws), ws.numel(), _stream()), "pgdvs_points_raster_bounded")
    return {"rgb": rgb, "mask": mask, "idx": idx, "zbuf": zbuf, "dist2": d2, "status": status}


def check_raster_status(status, what: str = "pgdvs_points_raster_bounded") -> None:
    """Host read of the bounded rasteriser's status word (synchronises): raises when rows were cut off."""
    if status is None:
        return
    for v in status.reshape(-1).tolist():
        if v == 1:
            raise PgdvsHipError(f"{what}: the device-side point count exceeds the row bound the workspace was sized for; "
                                "rows were cut off and the static image is not valid -- pass a larger bound")
        if v != 0:
            raise PgdvsHipError(f"{what}: the device-side point count is negative (the producer's error status); nothing was drawn")


def mesh_render(cam_tgt, keep, pcl, rgb, H: int, W: int, want_faces: bool = False):
    """A10 mesh variant -> dict(rgb[3,H,W], mask[H,W][, face[H,W] int32])."""
    cam = _req(cam_tgt, torch.float32, "cam_tgt")
    img = torch.empty((3, H, W), dtype=torch.float32, device=cam.device)
    mask = torch.empty((H, W), dtype=torch.float32, device=cam.device)
    face = torch.empty((H, W), dtype=torch.int32, device=cam.device) if want_faces else None
    lib = _lib.load()
    ws = _ws(lib.pgdvs_mesh_render_workspace_bytes(H, W), cam.device)
    check(lib.pgdvs_mesh_render(_ptr(cam), H, W, _ptr(_req(keep, torch.uint8, "keep")), _ptr(_req(pcl, torch.float32, "pcl")),
                                _ptr(_req(rgb, torch.float32, "rgb")), _ptr(img), _ptr(mask), _ptr(face), _ptr(ws), ws.numel(),
                                _stream()), "pgdvs_mesh_render")
    out = {"rgb": img, "mask": mask}
    if want_faces:
        out["face"] = face
    return out


def static_aggregate(rgbs, depths, dyn_masks, K3s, c2ws, capacity: int | None = None, return_xyz: bool = False):
    """rgbs[S,H,W,3] fp32 in [0,1]; depths[S,H,W]; dyn_masks[S,H,W] bool/uint8 (GPU tensors);
    K3s[S,3,3], c2ws[S,4,4] float64 numpy (host).  -> (cloud[capacity,6], count[int64 dev]) and, with
    ``return_xyz``, the packed coordinates [capacity,3] (``data["st_pcl_xyz"]`` for the renderer)."""
    r = _req(rgbs, torch.float32, "rgbs")
    d = _req(depths, torch.float32, "depths")
    m = _req(dyn_masks.contiguous().view(torch.uint8) if dyn_masks.dtype == torch.bool else dyn_masks, torch.uint8, "dyn_masks")
    S, H, W = d.shape
    K3 = np.ascontiguousarray(K3s, dtype=np.float64).reshape(S, 9)
    c2w = np.ascontiguousarray(c2ws, dtype=np.float64).reshape(S, 16)
    cap = int(capacity) if capacity is not None else S * H * W
    out = torch.empty((cap, 6), dtype=torch.float32, device=r.device)
    cnt = torch.empty(1, dtype=torch.int64, device=r.device)
    lib = _lib.load()
    ws = _ws(lib.pgdvs_static_aggregate_workspace_bytes(S, H, W, cap), r.device)
    if return_xyz:
        xyz = torch.empty((cap, 3), dtype=torch.float32, device=r.device)
        check(lib.pgdvs_static_aggregate_packed(
            _ptr(r), _ptr(d), _ptr(m), K3.ctypes.data_as(C.c_void_p), c2w.ctypes.data_as(C.c_void_p), S, H, W,
            _ptr(out), _ptr(xyz), cap, _ptr(cnt), _ptr(ws), ws.numel(), _stream()), "pgdvs_static_aggregate_packed")
        return out, cnt, xyz
    check(lib.pgdvs_static_aggregate(
        _ptr(r), _ptr(d), _ptr(m), K3.ctypes.data_as(C.c_void_p), c2w.ctypes.data_as(C.c_void_p), S, H, W,
        _ptr(out), cap, _ptr(cnt), _ptr(ws), ws.numel(), _stream()), "pgdvs_static_aggregate")
    return out, cnt


class ViewGeoState:
    """What ``view_geo_forward`` keeps between calls for ONE HIP stream: the workspace of the native call (reused, so a
    view costs no allocator round trips beyond its outputs) and the description struct.  Views in flight on different
    streams need different states (``PGDVSRenderer`` keeps one per stream it is called on)."""

    def __init__(self):
        self.desc = _lib.ViewGeoDesc()
        self.workspace = None
        self.ws_key = None
        self.ws_bytes = 0
        self.agg_key = None  # (shape, cameras) whose per-frame constants the workspace holds
        self.default_side_stream = None  # PGDVSRenderer._forward_native: the dynamic branch's stream when the caller names none


def view_geo_forward(state: ViewGeoState, *, H: int, W: int, flat_cam_tgt, flat_cam_src, time_src, time_tgt, rgb1, rgb2, depth1,
                     depth2, dyn_mask1, flow12, flow_occ, use_flow_consistency: bool, remove_outlier: bool, outlier_knn: int,
                     outlier_std_thres: float, alpha: float, noise=None, rng_state=None, st_pcl_rgb=None, st_pcl_xyz=None,
                     st_count=None, video=None, row_bound=None, radius: float, K: int, out_combined=None, side_stream=None):
    """ONE native call for the whole geometric per-view path (``pgdvs_view_geo_forward``): A12 (when ``video`` is given)
    + A9 + A2-A5 + A6-A8 + A11.  All tensors fp32 contiguous on the GPU (checked; no conversions are made here -- the
    caller falls back to the per-op path for anything else).
    ``video``: dict(rgbs[S,H,W,3], depths[S,H,W], dyn_masks[S,H,W] u8, K3s, c2ws (float64 numpy), capacity) -- the cloud
    is aggregated inside the call and returned (``st_pcl_rgb``, ``st_pcl_xyz``, ``st_pcl_rgb_count``).
    Returns a dict of planar images (see the keys below)."""
    lib = _lib.load()
    d = state.desc
    dev = rgb1.device
    P = H * W

    def ptr(t, name, dtype=torch.float32, numel=None):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            raise PgdvsHipError(f"view_geo_forward: {name} must be a contiguous {dtype} GPU tensor (got {t.dtype}, {t.device}, "
                                f"contiguous={t.is_contiguous()})")
        if numel is not None and t.numel() != numel:
            raise PgdvsHipError(f"view_geo_forward: {name} has {t.numel()} elements, expected {numel}")
        return t.data_ptr()

    d.H, d.W = H, W
    d.flat_cam_tgt = ptr(flat_cam_tgt, "flat_cam_tgt", numel=34)
    d.flat_cam_src = ptr(flat_cam_src, "flat_cam_src", numel=68)
    d.time_src = ptr(time_src, "time_src")
    d.time_tgt = ptr(time_tgt, "time_tgt")
    if time_src.numel() < 2 or time_tgt.numel() < 1:
        raise PgdvsHipError("view_geo_forward: time_src needs 2 entries, time_tgt 1")
    d.rgb1, d.rgb2 = ptr(rgb1, "rgb1", numel=3 * P), ptr(rgb2, "rgb2", numel=3 * P)
    d.depth1, d.depth2 = ptr(depth1, "depth1", numel=P), ptr(depth2, "depth2", numel=P)
    d.dyn_mask1 = ptr(dyn_mask1, "dyn_mask1", numel=P)
    d.flow12 = ptr(flow12, "flow12", numel=2 * P)
    d.flow_occ = ptr(flow_occ, "flow_occ", numel=P)
    d.use_flow_consistency, d.remove_outlier, d.outlier_knn = int(bool(use_flow_consistency)), int(bool(remove_outlier)), int(outlier_knn)
    d.outlier_std_thres, d.alpha = float(outlier_std_thres), float(alpha)
    d.noise = ptr(noise, "noise", numel=3 * P)
    d.rng_state = ptr(rng_state, "rng_state", torch.int64, 2) if noise is None else None
    out = {}
    keep_alive = None
    if video is not None:
        r, dp, m = video["rgbs"], video["depths"], video["dyn_masks"]
        S = dp.shape[0]
        cap = int(video.get("capacity") or S * P)
        K3 = np.ascontiguousarray(video["K3s"], dtype=np.float64).reshape(S, 9)
        c2w = np.ascontiguousarray(video["c2ws"], dtype=np.float64).reshape(S, 16)
        keep_alive = (K3, c2w)
        block = torch.empty(cap * 9, dtype=torch.float32, device=dev)  # cloud rows and packed coordinates in one block
        cloud, xyz = block[: cap * 6].view(cap, 6), block[cap * 6:].view(cap, 3)
        cnt = torch.empty(1, dtype=torch.int64, device=dev)
        d.agg_S = S
        d.agg_rgbs, d.agg_depths = ptr(r, "video.rgbs", numel=S * P * 3), ptr(dp, "video.depths", numel=S * P)
        d.agg_masks = ptr(m, "video.dyn_masks", torch.uint8, S * P)
        d.agg_K3s_host, d.agg_c2ws_host = K3.ctypes.data, c2w.ctypes.data
        d.agg_cloud_out, d.agg_xyz_out, d.agg_capacity, d.agg_count_out = cloud.data_ptr(), xyz.data_ptr(), cap, cnt.data_ptr()
        d.st_pcl_rgb = d.st_pcl_xyz = d.st_count_dev = None
        d.st_rows = 0
        rows = cap
        out.update(st_pcl_rgb=cloud, st_pcl_xyz=xyz, st_pcl_rgb_count=cnt)
        agg_key = (S, H, W, cap, K3.tobytes(), c2w.tobytes())
    else:
        agg_key = None
        d.agg_S = 0
        rows = st_pcl_rgb.shape[0]
        d.st_pcl_rgb = ptr(st_pcl_rgb, "st_pcl_rgb", numel=rows * 6) if rows else None
        d.st_pcl_xyz = ptr(st_pcl_xyz, "st_pcl_xyz", numel=rows * 3) if (st_pcl_xyz is not None and rows) else None
        d.st_rows = rows
        d.st_count_dev = ptr(st_count, "st_pcl_rgb_count", torch.int64, 1)
    d.row_bound = int(row_bound) if (row_bound is not None and (video is not None or st_count is not None)) else 0
    d.radius, d.K = float(radius), int(K)
    # outputs: one block for the images, one for the masks
    n_img = 4 if out_combined is not None else 5
    imgs = torch.empty((n_img, 3, H, W), dtype=torch.float32, device=dev)
    masks = torch.empty((2, H, W), dtype=torch.float32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    if out_combined is not None:
        if not (out_combined.is_cuda and out_combined.dtype == torch.float32 and out_combined.is_contiguous()
                and out_combined.numel() == 3 * P):
            raise PgdvsHipError(f"out_combined: expected a contiguous fp32 GPU tensor [3,{H},{W}], got {tuple(out_combined.shape)}")
        comb = out_combined.view(3, H, W)
    else:
        comb = imgs[4]
    d.static_rgb, d.render_dyn_rgb, d.combined_static, d.combined_dyn = (imgs[i].data_ptr() for i in range(4))
    d.combined = comb.data_ptr()
    d.static_mask, d.render_dyn_mask = masks[0].data_ptr(), masks[1].data_ptr()
    d.raster_status = status.data_ptr()
    d.side_stream = side_stream.cuda_stream if side_stream is not None else None
    # workspace: reused while the shape of the problem stays the same
    key = (H, W, d.agg_S, int(d.agg_capacity) if d.agg_S else rows, int(d.row_bound), d.radius, d.remove_outlier, dev)
    if state.ws_key != key:
        need = lib.pgdvs_view_geo_workspace_bytes(C.byref(d))
        if need < 0:
            check(int(need), "pgdvs_view_geo_workspace_bytes")
        if state.workspace is None or state.ws_bytes < need or state.workspace.device != dev:
            state.workspace = None  # (release first: two of these do not have to coexist)
            state.workspace = torch.empty(int(need), dtype=torch.uint8, device=dev)
            state.ws_bytes = int(need)
        state.ws_key = key
        state.agg_key = None  # (another layout, or a fresh block: nothing is cached in it)
    # the camera constants of the aggregation stay in the workspace between calls with the same video
    d.agg_params_cached = int(agg_key is not None and agg_key == state.agg_key)
    state.agg_key = None  # (until the call has gone through)
    check(lib.pgdvs_view_geo_forward(C.byref(d), state.workspace.data_ptr(), state.ws_bytes, _stream()), "pgdvs_view_geo_forward")
    del keep_alive
    state.agg_key = agg_key
    out.update(geo_static_rgb=imgs[0], geo_static_mask=masks[0], render_dyn_rgb=imgs[1], render_dyn_mask=masks[1],
               combined_rgb=comb, combined_rgb_static=imgs[2], combined_rgb_dyn=imgs[3], raster_status=status)
    return out


VIEW_COUNTER_NAMES = ("static_rows", "raster_list_entries", "raster_longest_tile_list", "raster_tiles_general_path_long_list",
                      "raster_tiles_general_path_equal_depths", "knn_queries", "knn_queries_to_ring_search",
                      "knn_queries_to_coarse_grid", "knn_queries_scanned_exhaustively", "agg_points_in_fp64_queue",
                      "agg_projections_in_reference_order")


def view_geo_counters(state: ViewGeoState) -> dict:
    """What the fast paths of the last ``view_geo_forward`` on ``state`` left to their slower exits
    (``pgdvs_view_geo_counters``; enqueued on the current stream, which must be the one the view ran on; synchronises)."""
    if state.workspace is None:
        raise PgdvsHipError("view_geo_counters: no view has been rendered with this state")
    out = torch.empty(12, dtype=torch.int64, device=state.workspace.device)
    check(_lib.load().pgdvs_view_geo_counters(C.byref(state.desc), state.workspace.data_ptr(), state.ws_bytes, out.data_ptr(), _stream()),
          "pgdvs_view_geo_counters")
    vals = out.tolist()
    return {k: int(vals[i]) for i, k in enumerate(VIEW_COUNTER_NAMES)}


def view_geo_host_stats():
    """(calls, seconds) spent inside ``pgdvs_view_geo_forward`` since the last call of this function (resets)."""
    calls, secs = C.c_int64(0), C.c_double(0.0)
    _lib.load().pgdvs_view_geo_host_stats(C.byref(calls), C.byref(secs))
    return int(calls.value), float(secs.value)


def eval_psnr_sums(pred_planar, gt_hwc, mask_hwc, want_images: bool = False, count_dev=None, status_dev=None):
    """The evaluator's per-view statistics in one pass (``pgdvs_eval_psnr_sums``): pred[3,H,W] raw render, gt[H,W,3] raw,
    mask[H,W,3] -> device float64[8] (sum d2, sum d2 m, sum d2 (1-m), count, sum m, sum (1-m), then ``count_dev`` -- int64[1],
    -1 when None -- and ``status_dev`` -- int32[1], 0 when None -- as doubles, so that one transfer brings everything back)
    and, with ``want_images``, the quantised prediction / ground truth [3,H,W]."""
    p = _req(pred_planar, torch.float32, "pred")
    g = _req(gt_hwc, torch.float32, "gt")
    m = _req(mask_hwc, torch.float32, "eval_mask")
    _, H, W = p.shape
    assert tuple(g.shape) == (H, W, 3) and tuple(m.shape) == (H, W, 3), (p.shape, g.shape, m.shape)
    lib = _lib.load()
    nws = int(lib.pgdvs_eval_psnr_workspace_bytes())
    buf = torch.empty(nws + 64, dtype=torch.uint8, device=p.device)  # partials, then the eight doubles
    sums = buf[nws:nws + 64].view(torch.float64)
    pq = torch.empty_like(p) if want_images else None
    gq = torch.empty_like(p) if want_images else None
    cd = _req(count_dev, torch.int64, "count_dev") if count_dev is not None else None
    sd = _req(status_dev, torch.int32, "status_dev") if status_dev is not None else None
    check(lib.pgdvs_eval_psnr_sums(_ptr(p), _ptr(g), _ptr(m), H, W, _ptr(pq), _ptr(gq), _ptr(cd), _ptr(sd), _ptr(sums), _ptr(buf), nws,
                                   _stream()), "pgdvs_eval_psnr_sums")
    return sums, pq, gq


_pinned_sums = {}


def read_back_rows(rows):
    """[n] device tensors of 8 float64 each -> [n][8] Python floats, through ONE pinned staging block per device (cached):
    asynchronous copies on the current stream and one event wait, instead of a pageable `.cpu()` per step (which stages
    through a fresh host block and synchronises the whole stream)."""
    dev = rows[0].device
    n = len(rows)
    ent = _pinned_sums.get(dev.index)
    if ent is None or ent[0].shape[0] < n:
        ent = (torch.empty((max(n, 8), 8), dtype=torch.float64).pin_memory(), torch.cuda.Event())
        _pinned_sums[dev.index] = ent
    host, ev = ent
    for i, r_ in enumerate(rows):
        host[i].copy_(r_, non_blocking=True)
    ev.record()
    ev.synchronize()
    return host[:n].tolist()


def checked_count(cnt, what: str) -> int:
    """Host read of a device-side count that doubles as a status word: negative = the kernel chain
    reported an internal error (e.g. ``agg_select``'s ordered-offset look-back gave up) and its
    output must not be used."""
    n = int(cnt.item())
    if n < 0:
        raise PgdvsHipError(f"{what}: device-side error flag set (count {n}); the output is not valid")
    return n


def combine(static_rgb, dyn_rgb, dyn_mask):
    """[B,3,H,W], [B,3,H,W], [B,1,H,W] -> combined, combined_static, combined_dyn."""
    st = _req(static_rgb, torch.float32, "static_rgb")
    dy = _req(dyn_rgb, torch.float32, "dyn_rgb")
    mk = _req(dyn_mask, torch.float32, "dyn_mask")
    B, _, H, W = st.shape
    outs = [torch.empty_like(st) for _ in range(3)]
    lib = _lib.load()
    for b in range(B):
        check(lib.pgdvs_combine(_ptr(st[b]), _ptr(dy[b]), _ptr(mk[b]), H * W, _ptr(outs[0][b]), _ptr(outs[1][b]),
                                _ptr(outs[2][b]), _stream()), "pgdvs_combine")
    return outs


def gnt_gather(ray_o, ray_d, depth_range, n_samples: int, inv_uniform: bool, cam_tgt, cams_src, src_rgbs, featmaps_cl,
               inv_masks=None, z_samples=None):
    """A13.  ray_o/ray_d[R,3], depth_range[1,2] or [R,2], cams_src[V,80], src_rgbs[V,H,W,3],
    featmaps_cl[V,hf,wf,C], inv_masks[V,H,W] or None -> dict of [R,S,V,*] tensors."""
    ro, rd = _req(ray_o, torch.float32, "ray_o"), _req(ray_d, torch.float32, "ray_d")
    dr = _req(depth_range, torch.float32, "depth_range").reshape(-1, 2)
    R, S = ro.shape[0], int(n_samples)
    zs = None
    if z_samples is not None:  # explicit sample depths [R,S] (fine pass)
        zs = _req(z_samples, torch.float32, "z_samples")
        assert tuple(zs.shape) == (R, S), (zs.shape, R, S)
    img = _req(src_rgbs, torch.float32, "src_rgbs")
    V, H, W, _ = img.shape
    fm = _req(featmaps_cl, torch.float32, "featmaps_cl")
    hf, wf, Cc = fm.shape[1], fm.shape[2], fm.shape[3]
    per_ray = dr.shape[0] != 1
    assert dr.shape[0] in (1, R), dr.shape
    im = _req(inv_masks, torch.float32, "inv_masks").reshape(V, H, W) if inv_masks is not None else None
    dev = ro.device
    e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    out = {"pts": e(R, S, 3), "z_vals": e(R, S), "rgb_feat": e(R, S, V, 3 + Cc), "ray_diff": e(R, S, V, 4),
           "mask_inbound": e(R, S, V, 1), "mask_invalid": e(R, S, V, 1), "mask": e(R, S, V, 1)}
    ct, cs = _req(cam_tgt, torch.float32, "cam_tgt"), _req(cams_src, torch.float32, "cams_src")
    check(_lib.load().pgdvs_gnt_gather(
        _ptr(ro), _ptr(rd), _ptr(dr), int(per_ray), _ptr(zs), R, S, int(bool(inv_uniform)), _ptr(ct), _ptr(cs), V, _ptr(img), H, W,
        _ptr(fm), hf, wf, Cc, _ptr(im), _ptr(out["pts"]), _ptr(out["z_vals"]), _ptr(out["rgb_feat"]), _ptr(out["ray_diff"]),
        _ptr(out["mask_inbound"]), _ptr(out["mask_invalid"]), _ptr(out["mask"]), _stream()), "pgdvs_gnt_gather")
    if im is None:
        out["mask_invalid"].zero_()
    return out


# ---- packed-weight caches of the fused GNT kernels ---------------------------------------
def _param_key(*modules):
    """(storage address, in-place version) of every parameter: changes on `.to(device)`, on
    `load_state_dict` (in-place copies bump `_version`) and on optimiser steps."""
    return tuple((p.data_ptr(), p._version) for m in modules for p in m.parameters())


def _packed(owner, build, *modules):
    """Packed copy of `modules`' parameters cached on `owner`, rebuilt whenever a parameter was moved or
    written since it was packed (a checkpoint loaded after a warm-up forward must not keep the old weights)."""
    key = _param_key(*(modules or (owner,)))
    hit = getattr(owner, "_pgdvs_packed", None)
    if hit is None or hit[0] != key:
        hit = (key, build())
        object.__setattr__(owner, "_pgdvs_packed", hit)  # plain attribute: never a submodule / buffer / state_dict entry
    return hit[1]


def needs_autograd(*modules_and_tensors) -> bool:
    """The fused kernels return tensors without autograd history: they may run only when nobody can ask
    for gradients through them (inference under no_grad, or nothing on the path requires grad)."""
    if not torch.is_grad_enabled():
        return False
    for x in modules_and_tensors:
        if isinstance(x, torch.Tensor):
            if x.requires_grad:
                return True
        elif x is not None and any(p.requires_grad for p in x.parameters()):
            return True
    return False


_fallback_seen = set()


# ---------------------------------------------------------------- library options (include/pgdvs_hip.h, "Options")
def set_option(name: str, value) -> None:
    """process-wide option of the library (read from the environment once, at load time; this is the only other way to
    change one).  Names: agg_ordered, agg_stage, gnt_fp32, raster_bound_density, knn_no_tpq, knn_stats."""
    check(_lib.load().pgdvs_option_set(name.encode(), float(value)), f"pgdvs_option_set({name})")


def get_option(name: str) -> float:
    v = _lib.load().pgdvs_option_get(name.encode())
    if v != v:
        raise PgdvsHipError(f"pgdvs_option_get: unknown option '{name}'")
    return v


class option:
    """``with ops.option("agg_ordered", 1): ...`` -- sets a library option for the block and restores it (tests, bench.py)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def gnt_product_path(fp32: bool):
    """the GNT kernels' product path for a block of code: exact bf16x3 products on the bf16 matrix instructions (default) or
    every product on the fp32 matrix instruction (csrc/gnt_view.hip)"""
    return option("gnt_fp32", 1 if fp32 else 0)


def gnt_fallback(kernel: str, why: str) -> None:
    """A CUDA tensor is about to take the torch branch of a GNT stage because the fused kernel does not
    cover its shape: say so once per (kernel, reason); raise instead under PGDVS_GNT_STRICT=1."""
    import os
    import warnings

    if not _GNT_VIEW_ENABLED:  # the fused kernels are switched off on purpose (tests: the torch statement as the reference half)
        return
    msg = f"pgdvs_amd: {kernel} does not cover {why}; this stage runs on torch/rocBLAS (results identical, slower)"
    if os.environ.get("PGDVS_GNT_STRICT", "0") not in ("", "0"):
        raise PgdvsHipError(msg)
    if (kernel, why) not in _fallback_seen:
        _fallback_seen.add((kernel, why))
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


# ---- GNT view-transformer contraction on MFMA (csrc/gnt_view.hip) ------------------------
def gnt_view_available(dim: int, n_views: int) -> bool:
    """True when the fused MFMA view-layer kernel handles this shape (width 64)."""
    lib = _lib.load()
    return hasattr(lib, "pgdvs_gnt_view_layer") and dim == 64 and 1 <= n_views <= 64 and _GNT_VIEW_ENABLED


_GNT_VIEW_ENABLED = True


def _featc(t):
    """csrc/gnt_view.hip featc: feature index of register t of a 32-row tile's lane (+ 4 h for the lane's half)"""
    return (t & 3) + 8 * ((t & 15) >> 2) + 32 * (t >> 4)


_ff_img_index = None


def ff_bf16x3_images(w1_t: torch.Tensor, w2_t: torch.Tensor) -> torch.Tensor:
    """The feed-forward block's weights (input-major ``w1_t[64, 256]``, ``w2_t[256, 64]``) as the lane-major bf16x3 images of
    ``gnt_ff_bf16x3_kernel`` (csrc/gnt_view.hip): every weight split EXACTLY into three bf16 pieces (hi = w & 0xffff0000,
    mid = (w - hi) & 0xffff0000, lo = w - hi - mid), arranged [half of the hidden units][piece][W1 part | W2 part], a part
    being [chunk][lane][8 bf16] in the order the MFMA's A operand takes them (see the kernel).  Returns float32[49152] (the
    bit patterns, two bf16 per word)."""
    global _ff_img_index
    import numpy as np

    if _ff_img_index is None:
        lane = np.arange(64)
        m, kg = lane & 31, lane >> 5
        j = np.arange(8)
        idx1 = np.zeros((2, 4, 4, 64, 8), np.int64)      # [half][mtl][c][lane][j] -> flat index into w1_t (in * 256 + hid)
        idx2 = np.zeros((2, 4, 2, 2, 64, 8), np.int64)   # [half][mtl][c'][ot][lane][j] -> flat index into w2_t (hid * 64 + out)
        for hf in range(2):
            for mtl in range(4):
                mt = 4 * hf + mtl
                for c in range(4):
                    t = 8 * c + j                                        # [8]
                    fin = _featc(t)[None, :] + 4 * kg[:, None]           # [64, 8]
                    idx1[hf, mtl, c] = fin * 256 + (32 * mt + m)[:, None]
                for c2 in range(2):
                    r = 8 * c2 + j
                    hid = ((r & 3) + 8 * (r >> 2))[None, :] + 4 * kg[:, None] + 32 * mt
                    for ot in range(2):
                        idx2[hf, mtl, c2, ot] = hid * 64 + (32 * ot + m)[:, None]
        _ff_img_index = (torch.from_numpy(idx1.reshape(2, -1)), torch.from_numpy(idx2.reshape(2, -1)))
    i1, i2 = (x_.to(w1_t.device) for x_ in _ff_img_index)

    def pieces(w):  # float32 [n] -> three int32 [n] holding the bf16 bit patterns (upper halves)
        w = w.detach().float().contiguous().reshape(-1)
        mask = torch.tensor(-65536, dtype=torch.int32, device=w.device)
        hi = (w.view(torch.int32) & mask).view(torch.float32)
        r1 = w - hi
        mid = (r1.view(torch.int32) & mask).view(torch.float32)
        lo = r1 - mid
        return [((x_.view(torch.int32) >> 16) & 0xFFFF) for x_ in (hi, mid, lo)]

    p1, p2 = pieces(w1_t), pieces(w2_t)
    halves = []
    for hf in range(2):
        for p in range(3):
            halves.append(torch.cat([p1[p][i1[hf]], p2[p][i2[hf]]]))  # 8192 + 8192 bf16
    bits = torch.cat(halves).to(torch.int32)  # [98304] 16-bit patterns
    words = (bits[0::2] | (bits[1::2] << 16)).to(torch.int32)
    return words.view(torch.float32)


def pack_view_layer(layer) -> torch.Tensor:
    """Pack the parameters of one view-transformer layer (Transformer2D) into the input-major
    layout of csrc/gnt_view.hip (VW_* offsets)."""
    a = layer.attn
    dev = a.q_fc.weight.device

    def pad_cols(w_t, cols):  # [in][out] -> [in][cols]
        out = torch.zeros((w_t.shape[0], cols), dtype=torch.float32, device=dev)
        out[:, : w_t.shape[1]] = w_t
        return out

    def pad_vec(b, n):
        out = torch.zeros(n, dtype=torch.float32, device=dev)
        out[: b.numel()] = b
        return out

    parts = [
        layer.attn_norm.weight, layer.attn_norm.bias, a.q_fc.weight.t(), a.k_fc.weight.t(), a.v_fc.weight.t(),
        pad_cols(a.pos_fc[0].weight.t(), 32), pad_vec(a.pos_fc[0].bias, 32), a.pos_fc[2].weight.t(), a.pos_fc[2].bias,
        pad_cols(a.attn_fc[0].weight.t(), 32), pad_vec(a.attn_fc[0].bias, 32), a.attn_fc[2].weight.t(), a.attn_fc[2].bias,
        a.out_fc.weight.t(), a.out_fc.bias, layer.ff_norm.weight, layer.ff_norm.bias, layer.ff.fc1.weight.t(),
        layer.ff.fc1.bias, layer.ff.fc2.weight.t(), layer.ff.fc2.bias,
        ff_bf16x3_images(layer.ff.fc1.weight.t(), layer.ff.fc2.weight.t()),
    ]
    packed = torch.cat([p.detach().float().contiguous().reshape(-1) for p in parts])
    assert packed.numel() == _lib.load().pgdvs_gnt_view_weight_floats(), packed.numel()
    return packed


def gnt_view_layer(layer, q, feat, ray_diff, valid, want_stats):
    """q[R,S,64], feat[R,S,V,64], ray_diff[R,S,V,4], valid[R,S,V] bool -> (q_out, stats or None)."""
    packed = _packed(layer, lambda: pack_view_layer(layer))
    R, S, V = feat.shape[0], feat.shape[1], feat.shape[2]
    N = R * S
    qi = _req(q, torch.float32, "q")
    ft = _req(feat, torch.float32, "feat")
    rd = _req(ray_diff, torch.float32, "ray_diff")
    vd = _req(valid.view(torch.uint8) if valid.dtype == torch.bool else valid, torch.uint8, "valid")
    out = torch.empty_like(qi)
    stats = torch.empty((N, 3), dtype=torch.float32, device=q.device) if want_stats else None
    check(_lib.load().pgdvs_gnt_view_layer(_ptr(packed), _ptr(qi), _ptr(ft), _ptr(rd), _ptr(vd), N, V, _ptr(out), _ptr(stats),
                                           _stream()), "pgdvs_gnt_view_layer")
    if not want_stats:
        return out, None
    st = stats.reshape(R, S, 3)
    return out, (st[..., 0], st[..., 1], st[..., 2])


def gnt_embed_available(mlp, cin: int) -> bool:
    """True when the fused rgbfeat_fc kernel handles this network (3+32 channels -> 64 -> 64)."""
    return (_GNT_VIEW_ENABLED and 32 < cin <= 36 and mlp[0].out_features == 64 and mlp[2].out_features == 64
            and mlp[0].in_features == cin)


def pack_embed(mlp) -> torch.Tensor:
    """rgbfeat_fc (Linear -> ReLU -> Linear) in the input-major layout of csrc/gnt_embed.hip."""
    cin = mlp[0].in_features
    rows = (cin + 3) // 4 * 4
    w1 = torch.zeros((rows, 64), dtype=torch.float32, device=mlp[0].weight.device)
    w1[:cin] = mlp[0].weight.detach().float().t()
    parts = [w1, mlp[0].bias, mlp[2].weight.t(), mlp[2].bias]
    packed = torch.cat([p.detach().float().contiguous().reshape(-1) for p in parts])
    assert packed.numel() == _lib.load().pgdvs_gnt_embed_weight_floats(cin), packed.numel()
    return packed


def gnt_embed(mlp, rgb_feat, want_std: bool):
    """rgb_feat[R,S,V,Cin] -> feat[R,S,V,64], q0[R,S,64], (std[R,S], std_normalized[R,S]) or None."""
    packed = _packed(mlp, lambda: pack_embed(mlp))
    x = _req(rgb_feat, torch.float32, "rgb_feat")
    R, S, V, cin = x.shape
    N = R * S
    feat = torch.empty((R, S, V, 64), dtype=torch.float32, device=x.device)
    q0 = torch.empty((R, S, 64), dtype=torch.float32, device=x.device)
    stats = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_std else None
    check(_lib.load().pgdvs_gnt_embed(_ptr(packed), _ptr(x), N, V, cin, _ptr(feat), _ptr(q0), _ptr(stats), _stream()),
          "pgdvs_gnt_embed")
    if not want_std:
        return feat, q0, None
    st = stats.reshape(R, S, 2)
    return feat, q0, (st[..., 0], st[..., 1])


def gnt_posfc_available(q_fcs, dim: int) -> bool:
    mlps = [m for m in q_fcs if not isinstance(m, torch.nn.Identity)]
    return (_GNT_VIEW_ENABLED and dim == 64 and len(mlps) > 0
            and all(m[0].out_features == 64 and m[2].in_features == 64 and m[2].out_features == 64 for m in mlps))


class GntPosFc:
    """Per-forward state of the even layers' positional re-embedding (csrc/gnt_embed.hip,
    pgdvs_gnt_posfc): the position / direction parts of every q_fc's first layer come from one
    GEMM each, the per-row part runs in the MFMA kernel."""

    def __init__(self, q_fcs, pe_pts, pe_view):
        """pe_pts[R,S,P], pe_view[R,P']"""
        self.ids = [i for i, m in enumerate(q_fcs) if not isinstance(m, torch.nn.Identity)]
        mlps = [q_fcs[i] for i in self.ids]
        P, Pv = pe_pts.shape[-1], pe_view.shape[-1]
        def build():
            wp = torch.cat([m[0].weight.detach().float()[:, 64:64 + P] for m in mlps], 0).t().contiguous()  # [P, 64 L]
            wv = torch.cat([m[0].weight.detach().float()[:, 64 + P:64 + P + Pv] for m in mlps], 0).t().contiguous()
            b1 = torch.cat([m[0].bias.detach().float() for m in mlps], 0)
            packed = [torch.cat([m[0].weight.detach().float()[:, :64].t().contiguous().reshape(-1),
                                 m[2].weight.detach().float().t().contiguous().reshape(-1),
                                 m[2].bias.detach().float()]) for m in mlps]
            return (wp, wv, b1, packed)

        wp, wv, b1, self.packed = _packed(q_fcs, build)
        R, S = pe_pts.shape[0], pe_pts.shape[1]
        self.R, self.S = R, S
        self.T = pe_pts.reshape(R * S, P).float() @ wp        # [N, 64 L]
        self.tv = torch.addmm(b1, pe_view.float(), wv)        # [R, 64 L]

    def __call__(self, layer_index, q):
        k = self.ids.index(layer_index)
        qi = _req(q, torch.float32, "q")
        out = torch.empty_like(qi)
        N = self.R * self.S
        check(_lib.load().pgdvs_gnt_posfc(_ptr(self.packed[k]), _ptr(qi), self.T.data_ptr() + 256 * k, self.T.shape[1],
                                          self.tv.data_ptr() + 256 * k, self.tv.shape[1], N, self.S, _ptr(out), _stream()),
              "pgdvs_gnt_posfc")
        return out


def gnt_head_available(norm, rgb_fc) -> bool:
    return (_GNT_VIEW_ENABLED and tuple(norm.normalized_shape) == (64,) and abs(norm.eps - 1e-5) < 1e-12
            and rgb_fc.in_features == 64 and rgb_fc.out_features == 3)


def gnt_head(norm, rgb_fc, q):
    """rgb_fc(norm(q).mean(dim=1)): q[R,S,64] -> [R,3]"""
    packed = _packed(rgb_fc, lambda: torch.cat([p.detach().float().contiguous().reshape(-1)
                                                for p in (norm.weight, norm.bias, rgb_fc.weight, rgb_fc.bias)]), norm, rgb_fc)
    qi = _req(q, torch.float32, "q")
    R, S, _ = qi.shape
    out = torch.empty((R, 3), dtype=torch.float32, device=q.device)
    check(_lib.load().pgdvs_gnt_head(_ptr(packed), _ptr(qi), R, S, _ptr(out), _stream()), "pgdvs_gnt_head")
    return out


def pack_ray_layer(layer) -> torch.Tensor:
    """Ray-transformer layer (Transformer) in the same packed layout; view-only regions stay 0."""
    a = layer.attn
    dev = a.q_fc.weight.device
    z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
    parts = [
        layer.attn_norm.weight, layer.attn_norm.bias, a.q_fc.weight.t(), a.k_fc.weight.t(), a.v_fc.weight.t(),
        z(128 + 32 + 512 + 64 + 2048 + 32 + 512 + 64),
        a.out_fc.weight.t(), a.out_fc.bias, layer.ff_norm.weight, layer.ff_norm.bias, layer.ff.fc1.weight.t(),
        layer.ff.fc1.bias, layer.ff.fc2.weight.t(), layer.ff.fc2.bias,
        ff_bf16x3_images(layer.ff.fc1.weight.t(), layer.ff.fc2.weight.t()),
    ]
    packed = torch.cat([p.detach().float().contiguous().reshape(-1) for p in parts])
    assert packed.numel() == _lib.load().pgdvs_gnt_view_weight_floats(), packed.numel()
    return packed


GNT_RAY_MAX_SAMPLES = 256  # K and V of one ray live in LDS (csrc/gnt_view.hip)


def gnt_ray_available(dim: int, n_samples: int, n_heads: int) -> bool:
    return _GNT_VIEW_ENABLED and dim == 64 and n_heads == 4 and 1 <= n_samples <= GNT_RAY_MAX_SAMPLES


def gnt_ray_layer(layer, q, want_attn: bool):
    """q[R,S,64] -> (q_out[R,S,64], weights[R,S] or None)."""
    packed = _packed(layer, lambda: pack_ray_layer(layer))
    qi = _req(q, torch.float32, "q")
    R, S, _ = qi.shape
    out = torch.empty_like(qi)
    w = torch.empty((R, S), dtype=torch.float32, device=q.device) if want_attn else None
    check(_lib.load().pgdvs_gnt_ray_layer(_ptr(packed), _ptr(qi), R, S, _ptr(out), _ptr(w), _stream()), "pgdvs_gnt_ray_layer")
    return out, w
