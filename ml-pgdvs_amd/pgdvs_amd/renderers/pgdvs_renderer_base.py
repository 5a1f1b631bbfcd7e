"""Mirror of ``pgdvs.renderers.pgdvs_renderer_base.PGDVSBaseRenderer``
(pgdvs/renderers/pgdvs_renderer_base.py:16-138) on the HIP kernels."""
import torch

from .. import ops
from ..utils import softsplat as softsplat_mod


class PGDVSBaseRenderer(torch.nn.Module):
    def get_batched_rays(self, *, device, batch_size, H, W, render_stride, intrinsics, c2w):
        """Same contract as the reference (:17-57): integer pixel centres, un-normalised
        directions; returns (rays_o[B*n,3], rays_d[B*n,3], uvs[B*n,2], batch_refs, (rh, rw))."""
        flat = torch.zeros((batch_size, 34), dtype=torch.float32, device=device)
        flat[:, 0], flat[:, 1] = H, W
        flat[:, 2:18] = intrinsics.reshape(batch_size, 16).to(device=device, dtype=torch.float32)
        flat[:, 18:34] = c2w.reshape(batch_size, 16).to(device=device, dtype=torch.float32)
        cams = ops.cam_prep(flat)
        ros, rds, uvs = [], [], []
        shape = None
        for b in range(batch_size):
            ro, rd, uv, shape = ops.get_rays(cams[b], H, W, render_stride)
            ros.append(ro)
            rds.append(rd)
            uvs.append(uv)
        n = shape[0] * shape[1]
        batch_refs = torch.arange(batch_size).reshape((batch_size, 1)).expand(-1, n).reshape(-1)
        return torch.cat(ros, 0), torch.cat(rds, 0), torch.cat(uvs, 0), batch_refs, shape

    def softsplat_img(self, *, rgb_src1, flow_src1_to_tgt, rgb_src2=None, flow_src1_to_src2=None,
                      softsplat_metric_src1_to_src2=None):
        """:59-89 -- metric = mean_c|rgb1 - backwarp(rgb2)| unless supplied; soft splat."""
        if softsplat_metric_src1_to_src2 is None:
            softsplat_metric_src1_to_src2 = ops.backwarp_l1(rgb_src1, rgb_src2, flow_src1_to_src2)
        a = self.softsplat_metric_abs_alpha
        splat = softsplat_mod.softsplat(
            tenIn=rgb_src1, tenFlow=flow_src1_to_tgt,
            tenMetric=(-a * softsplat_metric_src1_to_src2).clip(-a, a), strMode="soft")
        return splat, softsplat_metric_src1_to_src2
