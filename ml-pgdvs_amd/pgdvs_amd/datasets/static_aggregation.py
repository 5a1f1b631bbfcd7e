"""Static point-cloud aggregation across the source frames of a video -- the GPU statement
of ``NvidiaDynPureGeoEvaluationDataset._aggregate_static_pcl`` and
``_compute_pcl_proj_mask`` (pgdvs/datasets/nvidia_eval_pure_geo.py:183-277) on in-memory
frames (disk I/O stays with the caller)."""
import numpy as np
import torch

from .. import ops


def hwf_to_K(h, w, f, tgt_shape=None):
    """``_hwf_to_K(normalized=False)`` (pgdvs/datasets/nvidia_eval.py:1013-1035), float64;
    ``tgt_shape`` = (h, w) rescales the intrinsics to another image size."""
    K = np.eye(3)
    K[0, 0] = f
    K[1, 1] = f
    K[0, 2] = w / 2.0
    K[1, 2] = h / 2.0
    if tgt_shape is not None:
        K[0, :] = K[0, :] * tgt_shape[1] / w
        K[1, :] = K[1, :] * tgt_shape[0] / h
    return K


def aggregate_static_pcl(rgbs, depths, dyn_masks, K3s, c2ws, *, sync=True, capacity=None):
    """rgbs[S,H,W,3] fp32 [0,1], depths[S,H,W] fp32, dyn_masks[S,H,W] bool -- GPU tensors;
    K3s[S,3,3], c2ws[S,4,4] float64 (numpy).  Returns ``st_pcl_rgb[#pt,6]`` in the reference's
    point order.  ``sync=False`` returns (buffer[capacity,6], count_dev) without reading the
    count back, for pipelines that keep everything on the device."""
    buf, cnt = ops.static_aggregate(rgbs, depths, dyn_masks, K3s, c2ws, capacity=capacity)
    if not sync:
        return buf, cnt
    return buf[: ops.checked_count(cnt, "pgdvs_static_aggregate")]
