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

    _MAX_VIEW_STATES = 16  # per-stream workspaces of the native call kept alive (views in flight use one stream each)

    # -- one native call per view (include/pgdvs_hip.h: pgdvs_view_geo_forward) -----------------------------------
    def _native_view_ok(self, data, render_cfg):
        """The geometric path with softsplat, one batch item, full resolution, everything fp32 on the GPU: what the
        reference's benchmark configurations run (scripts/benchmark.sh).  Anything else takes the per-op path below."""
        import os

        if os.environ.get("PGDVS_NATIVE_VIEW", "1") in ("0", ""):
            return False
        if not isinstance(self.static_renderer, StaticGeoPointRenderer) or self.dyn_renderer.use_tracker:
            return False
        if render_cfg.dyn_render_type != "softsplat" or render_cfg.render_stride != 1 or render_cfg.st_pcl_remove_outlier:
            return False
        if data.get("_dyn_prepared", None) is not None:
            return False
        t = data["rgb_src_temporal"]
        if not t.is_cuda or t.shape[0] != 1 or t.shape[1] < 2:
            return False
        if not 1 <= int(render_cfg.st_render_pcl_pts_per_pixel) <= 8:
            return False
        if render_cfg.dyn_pcl_remove_outlier and not 1 <= int(render_cfg.dyn_pcl_outlier_knn) <= 63:
            return False
        if "_st_pcl_video" not in data and ("st_pcl_rgb" not in data or data["st_pcl_rgb"].shape[1] >= (1 << 31)):
            return False
        keys = ["rgb_src_temporal", "depth_src_temporal", "dyn_mask_src_temporal", "flow_fwd", "flat_cam_tgt",
                "flat_cam_src_temporal", "time_src_temporal", "time_tgt"]
        if render_cfg.dyn_render_use_flow_consistency:
            keys.append("flow_fwd_occ_mask")
        for k in keys:
            v = data.get(k, None)
            if not (isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                return False
        for k in ("st_pcl_rgb", "st_pcl_xyz", "static_noise"):
            v = data.get(k, None)
            if v is not None and not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                return False
        return data["flat_cam_src_temporal"].shape[1] == 2 or data["flat_cam_src_temporal"][0, :2].is_contiguous()

    def release_view_states(self, keep_streams=()) -> int:
        """forget the native call's per-stream workspaces (0.8-2.2 GB each at 1080p) except those of ``keep_streams``
        (torch.cuda.Stream objects); returns how many were dropped.  The caller synchronises first."""
        states = self.__dict__.get("_view_states", {})
        keep = {(s_.device.index, s_.cuda_stream) for s_ in keep_streams if s_ is not None}
        drop = [k for k in states if k not in keep]
        for k in drop:
            states.pop(k)
        return len(drop)

    def _forward_native(self, data, render_cfg):
        """PGDVSRenderer.forward (:84-178) as ONE C-ABI call; same output keys and values as the per-op path."""
        _, _, H, W, _ = data["rgb_src_temporal"].shape
        dev = data["rgb_src_temporal"].device
        states = self.__dict__.setdefault("_view_states", {})
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        st = states.pop(key, None)
        if st is None:
            st = ops.ViewGeoState()
            while len(states) >= self._MAX_VIEW_STATES:  # (a workspace is ~1 GB at 1080p: forget the least recently used stream's)
                states.pop(next(iter(states)))
        states[key] = st  # (most recently used last)
        noise = data.get("static_noise", None)
        rng_state = None if noise is not None else self.dyn_renderer.splat_rng_state(dev)
        occ = data.get("flow_fwd_occ_mask", None)
        video = data.get("_st_pcl_video", None)
        counts = data.get("st_pcl_rgb_count", None)
        xyz = data.get("st_pcl_xyz", None)
        out_comb = data.get("_combined_rgb_out", None)
        # the dynamic branch's geometry beside the static branch, on a second stream forked and joined INSIDE the native call:
        # the caller's stream sees one call, as with the per-op path below (self._side_stream).  A plain
        # ``model.forward(data)`` -- what pgdvs.engines' evaluator does per view (evaluator_pgdvs.py:36-54) -- gets one such
        # stream per stream it calls from (round 6: a view alone took 1.37 ms on one stream, 1.13 on two); callers that
        # manage their own pass it as data["_side_stream"], or False for none.
        side = data.get("_side_stream", None)
        if side is None:
            side = getattr(st, "default_side_stream", None)
            if side is None:
                from ..runtime import stream_on_other_queue

                side = st.default_side_stream = stream_on_other_queue(torch.cuda.current_stream(dev))
        elif side is False:
            side = None
        r = ops.view_geo_forward(
            st, H=H, W=W, flat_cam_tgt=data["flat_cam_tgt"][0], flat_cam_src=data["flat_cam_src_temporal"][0, :2],
            time_src=data["time_src_temporal"][0], time_tgt=data["time_tgt"][0],
            rgb1=data["rgb_src_temporal"][0, 0], rgb2=data["rgb_src_temporal"][0, 1],
            depth1=data["depth_src_temporal"][0, 0], depth2=data["depth_src_temporal"][0, 1],
            dyn_mask1=data["dyn_mask_src_temporal"][0, 0], flow12=data["flow_fwd"][0],
            flow_occ=occ[0] if (occ is not None and occ.is_cuda and occ.dtype == torch.float32) else None,
            use_flow_consistency=render_cfg.dyn_render_use_flow_consistency,
            remove_outlier=render_cfg.dyn_pcl_remove_outlier, outlier_knn=render_cfg.dyn_pcl_outlier_knn,
            outlier_std_thres=render_cfg.dyn_pcl_outlier_std_thres, alpha=self.softsplat_metric_abs_alpha,
            noise=noise[0] if noise is not None else None, rng_state=rng_state,
            st_pcl_rgb=None if video is not None else data["st_pcl_rgb"][0],
            st_pcl_xyz=None if (video is not None or xyz is None) else xyz[0],
            st_count=None if (video is not None or counts is None) else counts.reshape(-1)[0:1],
            video=video, row_bound=data.get("st_pcl_rgb_row_bound", None),
            radius=render_cfg.st_render_pcl_pt_radius, K=render_cfg.st_render_pcl_pts_per_pixel,
            out_combined=out_comb[0] if out_comb is not None else None, side_stream=side)
        dyn_rgb, dyn_mask = r["render_dyn_rgb"][None], r["render_dyn_mask"][None, None]
        ret = {
            "geo_static_rgb": r["geo_static_rgb"][None], "geo_static_mask": r["geo_static_mask"][None, None],
            "geo_static_raster_status": r["raster_status"],
            "render_dyn_rgb": dyn_rgb, "render_dyn_mask": dyn_mask,
            "render_dyn_temporal_closest_rgb": dyn_rgb, "render_dyn_temporal_closest_mask": dyn_mask,
            "render_dyn_temporal_track_rgb": self.dyn_renderer._zeros_like(dyn_rgb),
            "render_dyn_temporal_track_mask": self.dyn_renderer._zeros_like(dyn_mask),
            "combined_rgb": out_comb if out_comb is not None else r["combined_rgb"][None],
            "combined_rgb_static": r["combined_rgb_static"][None], "combined_rgb_dyn": r["combined_rgb_dyn"][None],
        }
        if video is not None:  # the cloud the call aggregated, in the data dict's shapes
            ret["st_pcl_rgb"], ret["st_pcl_xyz"], ret["st_pcl_rgb_count"] = r["st_pcl_rgb"][None], r["st_pcl_xyz"][None], r["st_pcl_rgb_count"]
        return ret

    def view_counters(self, stream=None):
        """device counters of the last native view rendered on ``stream`` (default: the current one): list lengths and
        the number of tiles / queries / points that left a fast path (``ops.view_geo_counters``); synchronises"""
        dev = torch.cuda.current_device()
        s = stream if stream is not None else torch.cuda.current_stream()
        st = self.__dict__.get("_view_states", {}).get((s.device.index if hasattr(s, "device") else dev, s.cuda_stream))
        if st is None:
            raise ops.PgdvsHipError("view_counters: no native view has been rendered on this stream")
        with torch.cuda.stream(s):
            return ops.view_geo_counters(st)

    def forward(self, data, render_cfg={}, disable_tqdm=False, for_debug=False):
        if not for_debug and self._native_view_ok(data, render_cfg):
            return self._forward_native(data, render_cfg)
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
