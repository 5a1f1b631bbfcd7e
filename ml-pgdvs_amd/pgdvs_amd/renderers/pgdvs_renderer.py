"""Mirror of ``pgdvs.renderers.pgdvs_renderer.PGDVSRenderer``
(pgdvs/renderers/pgdvs_renderer.py:26-481): same constructor, same ``forward(data,
render_cfg, disable_tqdm, for_debug) -> dict`` contract and output keys, so
``pgdvs.engines`` can instantiate it through Hydra by pointing ``_target_`` at this
class."""
import torch

from .. import ops
from ..instantiate import instantiate
from ..models.gnt.renderer import BaseRenderer as GNTRenderer
from .pgdvs_renderer_base import PGDVSBaseRenderer
from .pgdvs_renderer_dyn import PGDVSDynamicRenderer
from .st_geo_renderer import StaticGeoPointRenderer


def disabled_train(self, mode=True):
    """pgdvs/utils/training.py disabled_train: keep the static renderer in eval mode."""
    return self


class PGDVSRenderer(PGDVSBaseRenderer):
    def __init__(self, cfg, *, render_cfg, flag_debug=False, train_static_renderer=False,
                 softsplat_metric_abs_alpha=100.0, local_rank=0):
        super().__init__()
        self.cfg = cfg
        self.flag_debug = flag_debug
        self.static_renderer = None
        self.refine_renderer = None
        self._side_stream = None

        if self.cfg.static_renderer._target_ is not None:
            self.static_renderer = instantiate(self.cfg.static_renderer)
            if not train_static_renderer:
                self.static_renderer = self.static_renderer.eval()
                self.static_renderer.train = disabled_train.__get__(self.static_renderer)

        self.softsplat_metric_abs_alpha = softsplat_metric_abs_alpha
        assert self.softsplat_metric_abs_alpha >= 0, f"{self.softsplat_metric_abs_alpha}"
        assert render_cfg.dyn_render_track_temporal in ["none", "no_tgt"], f"{render_cfg.dyn_render_track_temporal}"

        # :62-72
        if render_cfg.dyn_render_track_temporal == "no_tgt":
            from .pgdvs_renderer_dyn_track import PGDVSDynamicTrackRenderer

            dyn_renderer_cls = PGDVSDynamicTrackRenderer
        else:
            dyn_renderer_cls = PGDVSDynamicRenderer
        self.dyn_renderer = dyn_renderer_cls(
            cfg=cfg, softsplat_metric_abs_alpha=softsplat_metric_abs_alpha,
            proj_func=self.static_renderer.projector.compute_projections, local_rank=local_rank,
            use_tracker=render_cfg.dyn_render_track_temporal == "no_tgt")

    def forward(self, data, render_cfg={}, disable_tqdm=False, for_debug=False):
        n_b, _, orig_h, orig_w, _ = data["rgb_src_temporal"].shape
        ray_batch = self.prepare_ray_batch(data=data, B=n_b, H=orig_h, W=orig_w,
                                           render_stride=render_cfg.render_stride, render_cfg=render_cfg)
        ret_dict = {}
        # The dynamic branch's geometry (unproject, flow warp, kNN filter, projection) does not
        # depend on the static branch: enqueue it on a side HIP stream so it overlaps with the
        # static renderer; the splat + composite joins both.  Callers may pass an already
        # started preparation through data["_dyn_prepared"].
        prepared = data.get("_dyn_prepared", None)
        if prepared is None and data["rgb_src_temporal"].is_cuda \
                and not (render_cfg.pure_gnt or render_cfg.pure_gnt_with_dyn_mask):
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=data["rgb_src_temporal"].device)
            prepared = self.dyn_renderer.prepare(data, render_cfg, stream=self._side_stream)
        if isinstance(self.static_renderer, GNTRenderer):
            if "rgb_gnt" in data:
                static_rgb = data["rgb_gnt"].permute(0, 3, 1, 2)
                ret_dict["static_coarse_rgb"] = static_rgb
            else:
                static_rgb, st_ret_dict = self.forward_st_gnt(data=data, ray_batch=ray_batch, render_cfg=render_cfg,
                                                              disable_tqdm=disable_tqdm)
                ret_dict.update(st_ret_dict)
            if render_cfg.pure_gnt or render_cfg.pure_gnt_with_dyn_mask:
                ret_dict["combined_rgb"] = static_rgb
                return ret_dict
        elif isinstance(self.static_renderer, StaticGeoPointRenderer):
            # (the dynamic branch's preparation already holds the target cameras' blocks: same stream -> no second launch)
            cams_tgt = prepared["cams_tgt"] if (prepared is not None and prepared.get("stream", None) is None) else None
            static_rgb, st_ret_dict = self.forward_st_geo(data=data, ray_batch=ray_batch, render_cfg=render_cfg,
                                                          cams_tgt=cams_tgt)
            ret_dict.update(st_ret_dict)
        else:
            raise TypeError(type(self.static_renderer))

        render_dyn_rgb, render_dyn_mask, render_dyn_info = self.dyn_renderer(
            data, ray_batch, render_cfg, for_debug=for_debug, disable_tqdm=disable_tqdm, static_rgb=static_rgb,
            prepared=prepared)

        ret_dict["render_dyn_rgb"] = render_dyn_rgb
        ret_dict["render_dyn_mask"] = render_dyn_mask
        ret_dict["render_dyn_temporal_closest_rgb"] = render_dyn_info["temporal_closest_rgb"]
        ret_dict["render_dyn_temporal_closest_mask"] = render_dyn_info["temporal_closest_mask"]
        ret_dict["render_dyn_temporal_track_rgb"] = render_dyn_info["temporal_track_rgb"]
        ret_dict["render_dyn_temporal_track_mask"] = render_dyn_info["temporal_track_mask"]

        # combine static and dynamic (:169-178); fused into the splat epilogue when possible
        if "combined_rgb" in render_dyn_info:
            combined_rgb = render_dyn_info["combined_rgb"]
            combined_rgb_static = render_dyn_info["combined_rgb_static"]
            combined_rgb_dyn = render_dyn_info["combined_rgb_dyn"]
        else:
            combined_rgb, combined_rgb_static, combined_rgb_dyn = ops.combine(static_rgb, render_dyn_rgb, render_dyn_mask)
        ret_dict["combined_rgb"] = combined_rgb
        ret_dict["combined_rgb_static"] = combined_rgb_static
        ret_dict["combined_rgb_dyn"] = combined_rgb_dyn
        return ret_dict

    def forward_st_geo(self, *, data, ray_batch, render_cfg, cams_tgt=None):
        """:182-201"""
        static_rgb, static_mask = [], []
        counts = data.get("st_pcl_rgb_count", None)  # optional device counts [B] (int64)
        xyz = data.get("st_pcl_xyz", None)  # optional packed coordinates [B,#pt,3] (ops.static_aggregate(return_xyz=True))
        # optional row bound (int) for the rasteriser's workspace when the cloud buffer is capacity-sized with a device
        # count: the status words come back as ret["geo_static_raster_status"] (harness.eval_step checks them)
        bound = data.get("st_pcl_rgb_row_bound", None)
        status = []
        for i_b in range(data["flat_cam_tgt"].shape[0]):
            tmp_rgb, tmp_mask = self.static_renderer(
                tgt_h=ray_batch["render_h"], tgt_w=ray_batch["render_w"], flat_tgt_cam=data["flat_cam_tgt"][i_b],
                st_pcl_rgb=data["st_pcl_rgb"][i_b], render_cfg=render_cfg,
                n_points_dev=None if counts is None else counts[i_b:i_b + 1], planar=True,
                st_pcl_xyz=None if xyz is None else xyz[i_b], row_bound=bound, status_out=status,
                cam_block=None if cams_tgt is None else cams_tgt[i_b])
            static_rgb.append(tmp_rgb)
            static_mask.append(tmp_mask)
        if len(static_rgb) == 1:  # a view, not a 25 MB copy
            ret_dict = {"geo_static_rgb": static_rgb[0][None], "geo_static_mask": static_mask[0][None]}
        else:
            ret_dict = {"geo_static_rgb": torch.stack(static_rgb, 0), "geo_static_mask": torch.stack(static_mask, 0)}
        if status:
            ret_dict["geo_static_raster_status"] = status[0] if len(status) == 1 else torch.cat(status)
        return ret_dict["geo_static_rgb"], ret_dict

    def forward_st_gnt(self, *, data, ray_batch, render_cfg, disable_tqdm=True):
        """:203-352 -- run the GNT static renderer and unpack its coarse outputs."""
        if render_cfg.pure_gnt:
            assert not render_cfg.gnt_use_dyn_mask
            assert not render_cfg.gnt_use_masked_spatial_src
        if render_cfg.pure_gnt_with_dyn_mask:
            assert render_cfg.gnt_use_dyn_mask
            assert not render_cfg.gnt_use_masked_spatial_src
        n_src_spatial = data["rgb_src_temporal"].shape[1]  # sic: the temporal count, as upstream (:212)
        static_ret = self.static_renderer(
            ray_batch=ray_batch, chunk_size=render_cfg.chunk_size, inv_uniform=render_cfg.sample_inv_uniform,
            n_coarse_samples_per_ray=render_cfg.n_coarse_samples_per_ray,
            n_fine_samples_per_ray=render_cfg.n_fine_samples_per_ray, use_dyn_mask=render_cfg.gnt_use_dyn_mask,
            flag_deterministic=True, render_stride=render_cfg.render_stride, ret_view_entropy=True, ret_view_std=True,
            disable_tqdm=disable_tqdm)
        oc = static_ret["outputs_coarse"]
        ret = {}
        for k in ("rgb", "depth", "view_entropy", "view_std", "view_std_normalized", "inbound_cnt"):
            ret[f"static_coarse_{k}"] = oc[k].permute(0, 3, 1, 2)
        ret["static_coarse_oob_mask"] = (
            ret["static_coarse_inbound_cnt"] < (render_cfg.mask_oob_n_proj_thres / n_src_spatial)).float()
        if render_cfg.gnt_use_dyn_mask:
            ret["static_coarse_dyn_cnt"] = oc["dyn_cnt"].permute(0, 3, 1, 2)
            ret["static_coarse_dyn_mask_any"] = (ret["static_coarse_dyn_cnt"] > 0.0).float()
            ret["static_coarse_dyn_mask_all"] = (ret["static_coarse_dyn_cnt"] == 1.0).float()
            ret["static_coarse_dyn_mask_thres"] = (
                ret["static_coarse_dyn_cnt"] >= (render_cfg.mask_invalid_n_proj_thres / n_src_spatial)).float()
        return ret["static_coarse_rgb"], ret

    def prepare_ray_batch(self, *, data, B, H, W, render_stride, render_cfg):
        """:354-417.  Target rays are only materialised for the GNT network; the geometric
        and rgb_gnt paths need just the render size."""
        render_h = (H + render_stride - 1) // render_stride
        render_w = (W + render_stride - 1) // render_stride
        ret = {
            "camera": data["flat_cam_tgt"], "rgb": data.get("rgb_tgt", None), "raw_h": H, "raw_w": W,
            "render_h": render_h, "render_w": render_w, "render_stride": render_stride,
        }
        if isinstance(self.static_renderer, GNTRenderer) and "rgb_gnt" not in data:
            tgt_K = data["flat_cam_tgt"][:, 2:18].reshape((B, 4, 4))
            tgt_c2w = data["flat_cam_tgt"][:, 18:34].reshape((B, 4, 4))
            ro, rd, uvs, refs, _ = self.get_batched_rays(
                device=data["rgb_src_temporal"].device, batch_size=B, H=H, W=W, render_stride=render_stride,
                intrinsics=tgt_K, c2w=tgt_c2w)
            src_rgbs = data["static_rgb_src_spatial"] if render_cfg.gnt_use_masked_spatial_src else data["rgb_src_spatial"]
            if data["depth_range"].ndim == 4:
                depth_range = data["depth_range"][:, ::render_stride, ::render_stride, :].reshape((-1, 2))
                per_ray = True
            elif data["depth_range"].ndim == 2:
                depth_range, per_ray = data["depth_range"], False
            else:
                raise ValueError(data["depth_range"].shape)
            ret.update({
                "ray_o": ro, "ray_d": rd, "batch_refs": refs, "view_uv": uvs, "depth_range": depth_range,
                "depth_range_per_ray": per_ray, "src_rgbs": src_rgbs, "src_invalid_masks": data["dyn_mask_src_spatial"],
                "src_cameras": data["flat_cam_src_spatial"]})
        return ret
