"""SECOND, independent restatement of the pytorch3d 0.7.4 boundary of row A9 (test infrastructure only).

pytorch3d is an un-vendored dependency of the reference (README.md:38, `conda install pytorch3d=0.7.4`)
and cannot be obtained in the build container (no wheel, no source, no network), so neither
`rasterize_points_cpu.cpp` nor the camera classes can be compiled or imported: **parity of A9 stays
unpinned**.  What can be bounded is how much the unknowable part matters.  oracle/pgdvs_oracle.c
restates the pipeline in pixel-friendly closed form; this file restates it the way pytorch3d itself
computes it, object by object, so that the two can be compared:

    cameras_from_opencv_projection   (pgdvs/utils/pytorch3d_utils.py:5-47 is a copy of it;
                                      call sites st_geo_renderer.py:86-88, pgdvs_renderer_dyn.py:685-690)
      R' = R^T with x,y columns negated, T' = t with x,y negated, f' = f / s, p0' = -(pp - wh/2) / s
    PointsRasterizer.transform
      pts_view = (Rotate(R').compose(Translate(T'))).transform_points(pts_world)
      pts_ndc  = (K'^T composed with the identity NDC transform).transform_points(pts_view)
      pts_ndc[..., 2] = pts_view[..., 2]
    Transform3d.transform_points: [x y z 1] @ M (one 4-term dot product per output, `torch.bmm`),
      then division by the 4th component
    RasterizePointsNaiveCpu: pixel centres from PixToNonSquareNdc with reversed indices, skip z < 0,
      strict dist2 < r^2, a max-heap of (z, idx, dist2) tuples trimmed to K
    NormWeightedCompositor: w = 1 - dist2 / r^2, out = sum w f / max(sum w, 1e-4)

The one thing reading cannot settle is the float32 rounding inside `torch.bmm` and `torch.inverse`:
BLAS kernels accumulate the 4 products in index order but may or may not fuse multiply and add, and
differ between the CPU (MKL / OpenBLAS) and the GPU (cuBLAS, FMA; nvcc also contracts
`dx*dx + dy*dy` in the CUDA rasteriser).  `flavour` selects the accumulation: "seq" rounds every
product and every sum (what pgdvs_oracle.c does), "fma" fuses each multiply-add (emulated through
float64: the product of two float32 is exact there).  tests/test_p3d_second.py measures how often
the two flavours -- i.e. the reference's own backends -- disagree on the z-buffer index.
"""
from __future__ import annotations

import heapq

import numpy as np

F = np.float32


def _dot4(rows, M, flavour):
    """rows[N,4] @ M[4,4] in float32, accumulating k = 0..3 in order."""
    rows = np.asarray(rows, F)
    M = np.asarray(M, F)
    out = np.empty((rows.shape[0], 4), F)
    for j in range(4):
        if flavour == "seq":
            acc = rows[:, 0] * M[0, j]
            for k in (1, 2, 3):
                acc = (acc + rows[:, k] * M[k, j]).astype(F)
        elif flavour == "fma":
            acc = (rows[:, 0].astype(np.float64) * np.float64(M[0, j])).astype(F)
            for k in (1, 2, 3):
                acc = (rows[:, k].astype(np.float64) * np.float64(M[k, j]) + acc.astype(np.float64)).astype(F)
        else:
            raise ValueError(flavour)
        out[:, j] = acc
    return out


class Transform3d:
    """row-vector convention, matrices composed left to right (pytorch3d/transforms/transform3d.py)"""

    def __init__(self, matrix=None):
        self._matrix = np.eye(4, dtype=F) if matrix is None else np.asarray(matrix, F).reshape(4, 4)
        self._transforms = []

    def compose(self, *others):
        out = Transform3d(self._matrix.copy())
        out._transforms = self._transforms + list(others)
        return out

    def get_matrix(self, flavour="seq"):
        m = self._matrix.copy()
        for other in self._transforms:
            m = _dot4(m, other.get_matrix(flavour), flavour)
        return m

    def transform_points(self, points, flavour="seq"):
        pts = np.asarray(points, F).reshape(-1, 3)
        hom = np.concatenate([pts, np.ones((pts.shape[0], 1), F)], axis=1)
        out = _dot4(hom, self.get_matrix(flavour), flavour)
        return (out[:, :3] / out[:, 3:]).astype(F)


class Rotate(Transform3d):
    def __init__(self, R):
        m = np.eye(4, dtype=F)
        m[:3, :3] = np.asarray(R, F)
        super().__init__(m)


class Translate(Transform3d):
    def __init__(self, T):
        m = np.eye(4, dtype=F)
        m[3, :3] = np.asarray(T, F)
        super().__init__(m)


class PerspectiveCameras:
    """in_ndc=True cameras as cameras_from_opencv_projection builds them"""

    def __init__(self, R, T, focal_length, principal_point):
        self.R, self.T = np.asarray(R, F), np.asarray(T, F)
        self.focal_length, self.principal_point = np.asarray(focal_length, F), np.asarray(principal_point, F)

    def get_world_to_view_transform(self):
        return Rotate(self.R).compose(Translate(self.T))

    def get_projection_transform(self):
        fx, fy = self.focal_length
        px, py = self.principal_point
        K = np.array([[fx, 0, px, 0], [0, fy, py, 0], [0, 0, 0, 1], [0, 0, 1, 0]], F)  # _get_sfm_calibration_matrix
        return Transform3d(K.T.copy())

    def get_ndc_camera_transform(self):
        return Transform3d()  # already in NDC


def cameras_from_opencv_projection(R, tvec, camera_matrix, image_size_hw):
    """R[3,3], tvec[3] = world-to-camera (OpenCV), camera_matrix[3,3], image_size (h, w)"""
    R, tvec, cm = np.asarray(R, F), np.asarray(tvec, F), np.asarray(camera_matrix, F)
    focal = np.array([cm[0, 0], cm[1, 1]], F)
    pp = cm[:2, 2].copy()
    wh = np.array([image_size_hw[1], image_size_hw[0]], F)
    scale = F(wh.min() / F(2.0))
    c0 = wh / F(2.0)
    focal_p3d = (focal / scale).astype(F)
    p0_p3d = (-(pp - c0) / scale).astype(F)
    R_p3d = R.T.copy()
    T_p3d = tvec.copy()
    R_p3d[:, :2] *= F(-1)
    T_p3d[:2] *= F(-1)
    return PerspectiveCameras(R_p3d, T_p3d, focal_p3d, p0_p3d)


def inverse_f32(c2w):
    """`torch.inverse(c2w)` on a float32 4x4: LAPACK single precision (getrf + getri), as torch's CPU path"""
    return np.linalg.inv(np.asarray(c2w, F)).astype(F)


def points_to_ndc(flat_cam_tgt, pts_world, flavour="seq", inverse="f32"):
    """st_geo_renderer.py:77-88 + PointsRasterizer.transform -> ndc[N,3] (x, y in NDC, z in view space)"""
    fc = np.asarray(flat_cam_tgt, F).reshape(-1)
    H, W = int(fc[0]), int(fc[1])
    K4, c2w = fc[2:18].reshape(4, 4), fc[18:34].reshape(4, 4)
    if inverse == "f32":
        w2c = inverse_f32(c2w)
    else:  # the closed-form oracle's choice: fp64 inverse rounded once
        w2c = np.linalg.inv(c2w.astype(np.float64)).astype(F)
    cams = cameras_from_opencv_projection(w2c[:3, :3], w2c[:3, 3], K4[:3, :3], (H, W))
    pts_view = cams.get_world_to_view_transform().transform_points(pts_world, flavour)
    proj = cams.get_projection_transform().compose(cams.get_ndc_camera_transform())
    ndc = proj.transform_points(pts_view, flavour)
    ndc[:, 2] = pts_view[:, 2]
    return ndc


def non_square_ndc_range(S1, S2):
    rng = F(2.0)
    if S1 > S2:
        rng = F(F(S1) * rng) / F(S2)  # "(S1 * range) / S2" of rasterization_utils
    return F(rng)


def pix_to_non_square_ndc(i, S1, S2):
    rng = non_square_ndc_range(S1, S2)
    offset = F(rng / F(2.0))
    return F(-offset + F(F(rng * F(i)) + offset) / F(S1))


def rasterize_points_naive(ndc, H, W, radius, K, fma_dist=False):
    """RasterizePointsNaiveCpu with its std::priority_queue of (z, idx, dist2) tuples (pure Python: small
    inputs only).  `fma_dist` = the CUDA flavour of dist2 (nvcc contracts dx*dx + dy*dy into one FMA)."""
    ndc = np.asarray(ndc, F)
    r2 = F(F(radius) * F(radius))
    idx = np.full((H, W, K), -1, np.int64)
    zbuf = np.full((H, W, K), -1, F)
    dist = np.full((H, W, K), -1, F)
    front = np.nonzero(~(ndc[:, 2] < 0))[0]
    for yi in range(H):
        yf = pix_to_non_square_ndc(H - 1 - yi, H, W)
        dy = (ndc[front, 1] - yf).astype(F)
        near_row = front[np.abs(dy) < np.sqrt(r2) * F(1.01) + F(1e-6)]
        for xi in range(W):
            xf = pix_to_non_square_ndc(W - 1 - xi, W, H)
            heap = []  # max-heap through negated keys
            for p in near_row:
                dx, dyy = F(ndc[p, 0] - xf), F(ndc[p, 1] - yf)
                if fma_dist:
                    d2 = F(np.float64(dx) * np.float64(dx) + np.float64(F(dyy * dyy)))
                else:
                    d2 = F(F(dx * dx) + F(dyy * dyy))
                if d2 < r2:
                    heapq.heappush(heap, (-float(ndc[p, 2]), -int(p), -float(d2)))
                    if len(heap) > K:
                        heapq.heappop(heap)  # drops the largest (z, idx, dist2)
            while heap:
                nz, nidx, nd = heapq.heappop(heap)
                i = len(heap)
                zbuf[yi, xi, i], idx[yi, xi, i], dist[yi, xi, i] = -nz, -nidx, -nd
    return idx, zbuf, dist


def norm_weighted_composite(idx, dist, radius, feat):
    """points/renderer.py: weights = 1 - dists2 / r^2; norm_weighted_sum_cpu.cpp"""
    H, W, K = idx.shape
    r2 = F(F(radius) * F(radius))
    out = np.zeros((H, W, feat.shape[1]), F)
    for yi in range(H):
        for xi in range(W):
            t = F(0)
            for k in range(K):
                if idx[yi, xi, k] < 0:
                    continue
                t = F(t + F(F(1) - F(dist[yi, xi, k] / r2)))
            t = max(t, F(1e-4))
            for k in range(K):
                n = idx[yi, xi, k]
                if n < 0:
                    continue
                w = F(F(1) - F(dist[yi, xi, k] / r2))
                out[yi, xi] = (out[yi, xi] + (w * feat[n]).astype(F) / t).astype(F)
    return out
