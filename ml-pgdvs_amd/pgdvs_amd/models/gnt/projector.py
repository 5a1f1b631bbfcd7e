"""Mirror of ``pgdvs.models.gnt.projector.Projector`` (pgdvs/models/gnt/projector.py:10-115):
the piece the dynamic renderer binds as ``proj_func`` (pgdvs_renderer.py:78)."""
import torch

from ... import ops


class Projector:
    def inbound(self, pixel_locations, h, w):
        # :14-27, closed bounds
        return ((pixel_locations[..., 0] <= w - 1.0) & (pixel_locations[..., 0] >= 0)
                & (pixel_locations[..., 1] <= h - 1.0) & (pixel_locations[..., 1] >= 0))

    def compute_projections(self, xyz, train_cameras):
        """:41-73.  xyz[#ray,#sample,3], train_cameras[#src,34] ->
        pixel_locations[#src,#ray,#sample,2], mask[#src,#ray,#sample] (mask is z>0 of the
        un-clamped projection, recomputed with torch since it is a by-product the dynamic
        renderer discards, pgdvs_renderer_dyn.py:470)."""
        if xyz.ndim != 3:
            raise AttributeError(xyz.shape)
        original_shape = xyz.shape[:2]
        cams = ops.cam_prep(train_cameras)
        pts = xyz.reshape(-1, 3).contiguous()
        uvs, masks = [], []
        for i in range(cams.shape[0]):
            uvs.append(ops.project_points(cams[i], pts))
            P = cams[i, 21:37].reshape(4, 4)
            z = pts @ P[2, :3] + P[2, 3]
            masks.append(z > 0)
        uv = torch.stack(uvs, 0).reshape((cams.shape[0],) + tuple(original_shape) + (2,))
        mask = torch.stack(masks, 0).reshape((cams.shape[0],) + tuple(original_shape))
        return uv, mask
