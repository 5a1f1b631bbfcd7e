"""Mirror of ``pgdvs.renderers.st_geo_renderer.StaticGeoPointRenderer``
(pgdvs/renderers/st_geo_renderer.py:20-122): optional statistical outlier removal, then
the point z-buffer rasteriser + norm-weighted compositor (pytorch3d semantics) in HIP."""
import torch

from .. import ops
from ..models.gnt.projector import Projector


class StaticGeoPointRenderer(torch.nn.Module):
    def __init__(self, model_cfg=None):
        super().__init__()
        self.projector = Projector()

    def forward(self, *, tgt_h, tgt_w, flat_tgt_cam, st_pcl_rgb, render_cfg, n_points_dev=None, planar=False,
                st_pcl_xyz=None, row_bound=None, status_out=None, cam_block=None):
        """st_pcl_rgb[#pt,6] (xyz,rgb).  Returns (mesh_img[H,W,3], mesh_mask[H,W,1]) like the
        reference, or planar ([3,H,W], [1,H,W]) with ``planar=True``.
        ``n_points_dev``: optional device int64 count (rows beyond it are ignored).
        ``st_pcl_xyz``: optional packed copy [#pt,3] of the coordinates (``ops.static_aggregate(...,
        return_xyz=True)``): the binning passes then read 12 instead of 24 bytes per point.
        ``row_bound``: with a device count, size the rasteriser's workspace for that many rows instead of the buffer's
        capacity; ``status_out`` (a list) then receives the device status word (``ops.check_raster_status``).
        ``cam_block``: the target camera's block if the caller already has it (``ops.cam_prep(flat_tgt_cam)``)."""
        assert st_pcl_rgb.ndim == 2, f"{st_pcl_rgb.shape}"
        cam = cam_block if cam_block is not None else ops.cam_prep(flat_tgt_cam)
        pts = st_pcl_rgb
        if st_pcl_xyz is not None and not render_cfg.st_pcl_remove_outlier:
            assert st_pcl_xyz.shape == (st_pcl_rgb.shape[0], 3), f"{st_pcl_xyz.shape}"
            r = ops.points_raster(
                st_pcl_xyz, st_pcl_rgb[:, 3:], cam, render_cfg.st_render_pcl_pt_radius,
                render_cfg.st_render_pcl_pts_per_pixel, tgt_h, tgt_w, n_points_dev=n_points_dev, rgb_planar=planar,
                row_bound=row_bound if n_points_dev is not None else None)
            if status_out is not None and "status" in r:
                status_out.append(r["status"])
            if planar:
                return r["rgb"], r["mask"][None]
            return r["rgb"], r["mask"][..., None]
        if render_cfg.st_pcl_remove_outlier:
            n = st_pcl_rgb.shape[0]
            if n_points_dev is not None:
                cnt = n_points_dev.to(torch.int32).reshape(1)
            else:
                cnt = torch.full((1,), n, dtype=torch.int32, device=st_pcl_rgb.device)
            xyz = st_pcl_rgb[:, :3].contiguous()
            avg = ops.knn_mean_dist(xyz, cnt, render_cfg.st_pcl_outlier_knn)
            _, flag = ops.outlier_flags(avg, cnt, render_cfg.st_pcl_outlier_std_thres, True)
            if n_points_dev is not None:
                flag = flag & (torch.arange(flag.numel(), device=flag.device) < cnt).to(torch.uint8)
            idx, cnt2 = ops.compact_u8(flag[:n])
            pts = ops.gather_rows(st_pcl_rgb, idx, cnt2)
            n_points_dev = cnt2.to(torch.int64)
        r = ops.points_raster(
            pts, pts[:, 3:], cam, render_cfg.st_render_pcl_pt_radius, render_cfg.st_render_pcl_pts_per_pixel,
            tgt_h, tgt_w, n_points_dev=n_points_dev, rgb_planar=planar,
            row_bound=row_bound if (n_points_dev is not None and not render_cfg.st_pcl_remove_outlier) else None)
        if status_out is not None and "status" in r:
            status_out.append(r["status"])
        if planar:
            return r["rgb"], r["mask"][None]
        return r["rgb"], r["mask"][..., None]
