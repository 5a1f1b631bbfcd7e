"""Mirror of ``pgdvs.renderers.pgdvs_renderer_dyn.PGDVSDynamicRenderer``
(pgdvs/renderers/pgdvs_renderer_dyn.py:28-724).

Same constructor / forward / compute_dyn_pcl contract.  The reference compacts the
masked pixels into variable-length point lists with boolean indexing (a host sync per
step); here the work stays dense over [H,W] with validity flags, the compact list for
the kNN filter is built on the device (ordered stream compaction) and no step reads a
value back to the host.
"""
import torch

from .. import ops
from .pgdvs_renderer_base import PGDVSBaseRenderer


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _tensors_of(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class PGDVSDynamicRenderer(PGDVSBaseRenderer):
    def __init__(self, *, cfg, softsplat_metric_abs_alpha=100.0, proj_func=None, local_rank=0, use_tracker=False):
        super().__init__()
        self.cfg = cfg
        self.softsplat_metric_abs_alpha = softsplat_metric_abs_alpha
        assert self.softsplat_metric_abs_alpha >= 0, f"{self.softsplat_metric_abs_alpha}"
        self.proj_func = proj_func
        self.tracker = None
        self.use_tracker = use_tracker
        if self.use_tracker:
            # :52-61.  The point trackers (TAPIR / CoTracker) are pretrained third-party
            # networks outside this build: any module with the reference's interface contract
            # (``tracker(frames=[N,H,W,3], query_points=[#pt,3]) -> tracks, visibles`` and a
            # ``query_chunk_size`` attribute) can be plugged in through cfg.tracker; without
            # one the tracks must be supplied in the data dict (see
            # PGDVSDynamicTrackRenderer.render_with_track).
            tracker_cfg = getattr(self.cfg, "tracker", None)
            if tracker_cfg is not None and getattr(tracker_cfg, "_target_", None) is not None:
                from ..instantiate import instantiate

                self.tracker = instantiate(tracker_cfg, ori_rgb_range=self.cfg.rgb_range, local_rank=local_rank)
                if isinstance(self.tracker, torch.nn.Module):
                    self.tracker = self.tracker.eval()
                self.track_chunk_size = self.tracker.query_chunk_size

    # -- A2..A5 -------------------------------------------------------------
    def compute_dyn_pcl(self, *, dyn_mask_1, rgb_1, depth_1, flow_12, flow_12_occ_mask, rgb_2, depth_2,
                        cam_1, cam_2, cam_tgt, times, render_cfg, need_points=False):
        """Dense statement of compute_dyn_pcl (:275-540).

        cam_* are camera blocks (ops.cam_prep); times = device tensor (t1, t2, t_tgt).
        Returns flow_1_to_tgt[2,H,W], valid_dyn_mask_1[H,W] and an info dict with
        device-side compact points when requested.
        """
        H, W = dyn_mask_1.shape[0], dyn_mask_1.shape[1]
        mask_eff, valid, pcl, rgbf = ops.dyn_warp(
            dyn_mask_1, flow_12_occ_mask, render_cfg.dyn_render_use_flow_consistency, flow_12, depth_1, depth_2,
            rgb_1, rgb_2, cam_1, cam_2, times)
        info = {"pcl_dense": pcl, "rgb_dense": rgbf, "valid": valid}
        # pytorch3d's kNN runs unconditionally upstream (:405-410) but its result is only
        # observable through the outlier flags (and the tracker, not built): skip it when
        # it cannot influence any output.
        if render_cfg.dyn_pcl_remove_outlier or self.use_tracker:  # the tracker row needs the threshold (:508)
            idx, cnt = ops.compact_u8(valid)
            pts = ops.gather_rows(pcl.reshape(-1, 3), idx, cnt)
            avg = ops.knn_mean_dist(pts, cnt, render_cfg.dyn_pcl_outlier_knn)
            thres, flag = ops.outlier_flags(avg, cnt, render_cfg.dyn_pcl_outlier_std_thres, render_cfg.dyn_pcl_remove_outlier)
            keep = ops.scatter_keep(idx, flag, cnt, H * W)
            info.update(pcl_nn_dist_thres=thres, avg_nn_dist=avg, n_valid=cnt)
        else:
            keep = valid.reshape(-1)
        info["keep"] = keep
        flow_1_to_tgt, valid_mask = ops.project_flow_dense(cam_tgt, pcl, keep, H, W)
        if need_points:
            idx2, cnt2 = ops.compact_u8(keep)
            info["pcl"] = ops.gather_rows(pcl.reshape(-1, 3), idx2, cnt2)
            info["pcl_rgbs"] = ops.gather_rows(rgbf.reshape(-1, 3), idx2, cnt2)
            info["n_pts"] = cnt2
        return flow_1_to_tgt, valid_mask, info

    def render_dyn_pcl(self, *, pcl, rgbs, n_pts, cam_tgt, H, W, render_cfg):
        """:671-724 -- point z-buffer + norm-weighted compositing of the dynamic cloud."""
        r = ops.points_raster(
            pcl, rgbs, cam_tgt, render_cfg.dyn_render_pcl_pt_radius, render_cfg.dyn_render_pcl_pts_per_pixel,
            H, W, n_points_dev=n_pts.to(torch.int64), rgb_planar=True)
        return r["rgb"], r["mask"]

    def render_dyn_mesh(self, *, keep, pcl_dense, rgb_dense, cam_tgt, H, W):
        """:542-669 -- pixel-grid triangulation of the kept source pixels, rendered with
        pytorch3d MeshRasterizer semantics (ops.mesh_render)."""
        r = ops.mesh_render(cam_tgt, keep, pcl_dense, rgb_dense, H, W)
        return r["rgb"], r["mask"]

    def render_with_track(self, *a, **kw):
        raise NotImplementedError  # :272-273; PGDVSDynamicTrackRenderer implements it

    # -- A2..A5 for a whole batch, optionally on a side stream ---------------------
    def prepare(self, data, render_cfg, stream=None):
        """Geometry half of forward(): camera blocks + compute_dyn_pcl for every batch item.
        It does not depend on the static branch, so PGDVSRenderer runs it on a side HIP stream
        while the static renderer works on the current one.  Returns an opaque dict that
        forward(..., prepared=...) consumes; the caller must make the consuming stream wait
        for ``prepared["stream"]`` (forward does)."""
        cur = torch.cuda.current_stream()
        if stream is not None:
            stream.wait_stream(cur)
        ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
        with ctx:
            n_b = data["rgb_src_temporal"].shape[0]
            dyn_type = render_cfg.dyn_render_type
            cams_src = ops.cam_prep(data["flat_cam_src_temporal"])  # [B,2,80]
            cams_tgt = ops.cam_prep(data["flat_cam_tgt"])  # [B,80]
            times = torch.cat([data["time_src_temporal"][:, :2].float(), data["time_tgt"][:, :1].float()], dim=1).contiguous()
            items = []
            for i_b in range(n_b):
                items.append(self.compute_dyn_pcl(
                    dyn_mask_1=data["dyn_mask_src_temporal"][i_b, 0, ..., 0],
                    rgb_1=data["rgb_src_temporal"][i_b, 0], depth_1=data["depth_src_temporal"][i_b, 0, ..., 0],
                    flow_12=data["flow_fwd"][i_b], flow_12_occ_mask=data["flow_fwd_occ_mask"][i_b, ..., 0],
                    rgb_2=data["rgb_src_temporal"][i_b, 1], depth_2=data["depth_src_temporal"][i_b, 1, ..., 0],
                    cam_1=cams_src[i_b, 0], cam_2=cams_src[i_b, 1], cam_tgt=cams_tgt[i_b], times=times[i_b],
                    render_cfg=render_cfg, need_points=(dyn_type == "pcl" or self.use_tracker)))
        return {"items": items, "cams_tgt": cams_tgt, "stream": stream}

    # -- A8 -----------------------------------------------------------------
    def forward(self, data, ray_batch, render_cfg, for_debug=False, disable_tqdm=False, static_rgb=None,
                prepared=None):
        """:63-257.  ``static_rgb`` [B,3,h,w] (optional) lets the splat epilogue also emit the
        static/dynamic composite of PGDVSRenderer.forward (:169-178) in the same pass.
        ``prepared``: result of prepare() (geometry already enqueued, possibly on a side stream)."""
        n_b, _, orig_h, orig_w, _ = data["rgb_src_temporal"].shape
        dev = data["rgb_src_temporal"].device
        assert self.cfg.rgb_range == "0_1", f"{self.cfg.rgb_range}"
        dyn_type = render_cfg.dyn_render_type
        if dyn_type not in ("softsplat", "pcl", "mesh"):
            raise ValueError(dyn_type)
        if prepared is None:
            prepared = self.prepare(data, render_cfg)
        if prepared["stream"] is not None:
            cur = torch.cuda.current_stream()  # (once: the call walks several Python layers, ~8 us)
            cur.wait_stream(prepared["stream"])
            if not torch.cuda.is_current_stream_capturing():  # (a captured graph owns its memory pool)
                for t in _tensors_of(prepared):
                    t.record_stream(cur)
        cams_tgt = prepared["cams_tgt"]

        render_h, render_w = ray_batch["render_h"], ray_batch["render_w"]
        same_res = (render_h == orig_h) and (render_w == orig_w)
        fuse_static = static_rgb is not None and same_res and dyn_type == "softsplat" and not self.use_tracker

        noise = rng_state = None
        if dyn_type == "softsplat":
            # torch.randn_like(rgb_src_1) upstream (:181): injectable for parity tests (``static_noise``); otherwise the
            # splat kernel draws the field itself, per forward, where it consumes it (only static pixels that land next
            # to dynamic content ever show it) -- seeded from torch's seed, one state per device, advanced by the kernel
            noise = data.get("static_noise", None)
            if noise is None:
                rng_state = self.splat_rng_state(dev)

        dyn_rgbs, dyn_masks, combs = [], [], []
        # optional caller-owned output [B,3,H,W] for combined_rgb (a slice of the caller's image stack): the
        # splat epilogue writes the composite there, so no copy follows the render
        out_comb = data.get("_combined_rgb_out", None)
        for i_b in range(n_b):
            flow_1_to_tgt, valid_mask, info = prepared["items"][i_b]
            if dyn_type == "softsplat":
                rgb, mask, c, cs, cd = ops.dyn_splat_composite(
                    data["rgb_src_temporal"][i_b, 0], data["rgb_src_temporal"][i_b, 1], data["flow_fwd"][i_b],
                    flow_1_to_tgt, valid_mask, noise[i_b] if noise is not None else None, self.softsplat_metric_abs_alpha,
                    static_rgb[i_b] if fuse_static else None,
                    out_combined=out_comb[i_b] if (fuse_static and out_comb is not None) else None, rng_state=rng_state)
                if fuse_static:
                    combs.append((c, cs, cd))
            elif dyn_type == "mesh":
                rgb, mask = self.render_dyn_mesh(keep=info["keep"], pcl_dense=info["pcl_dense"], rgb_dense=info["rgb_dense"],
                                                 cam_tgt=cams_tgt[i_b], H=orig_h, W=orig_w)
            else:
                rgb, mask = self.render_dyn_pcl(
                    pcl=info["pcl"], rgbs=info["pcl_rgbs"], n_pts=info["n_pts"], cam_tgt=cams_tgt[i_b],
                    H=orig_h, W=orig_w, render_cfg=render_cfg)
            dyn_rgbs.append(rgb)
            dyn_masks.append(mask[None])

        render_dyn_rgb = torch.stack(dyn_rgbs, 0) if n_b > 1 else dyn_rgbs[0][None]  # [B,3,H,W]
        render_dyn_mask = torch.stack(dyn_masks, 0) if n_b > 1 else dyn_masks[0][None]  # [B,1,H,W]

        if self.use_tracker:
            # :211-235 -- pixels the closest-frame rendering left empty are filled from the
            # tracker-window cloud
            infos = [it[2] for it in prepared["items"]]
            base_pcl_info = {k: [inf[k] for inf in infos] for k in ("pcl", "pcl_rgbs", "pcl_nn_dist_thres", "n_pts")}
            render_track_rgb, render_track_mask = self.render_with_track(
                data, render_cfg=render_cfg, base_pcl_info=base_pcl_info, for_debug=for_debug, disable_tqdm=disable_tqdm,
                cams_tgt=cams_tgt)
            mask_for_track = ((~(render_dyn_mask > 0)) & (render_track_mask > 0)).float()
            render_dyn_rgb_final = ops.combine(render_dyn_rgb, render_track_rgb, mask_for_track)[0]
            render_dyn_mask_final = ((render_dyn_mask > 0) | (render_track_mask > 0)).float()
        else:
            # no tracker: the track images are zeros, so the merge of :229-235 is the identity on
            # the {0,1}-valued closest-frame mask
            render_track_rgb = self._zeros_like(render_dyn_rgb)
            render_track_mask = self._zeros_like(render_dyn_mask)
            render_dyn_rgb_final, render_dyn_mask_final = render_dyn_rgb, render_dyn_mask

        if not same_res:
            if self.use_tracker:
                render_dyn_rgb_final, render_dyn_mask_final = self.resize_rgb_mask(
                    render_dyn_rgb_final, render_dyn_mask_final, render_h, render_w)
            render_dyn_rgb, render_dyn_mask = self.resize_rgb_mask(render_dyn_rgb, render_dyn_mask, render_h, render_w)
            render_track_rgb, render_track_mask = self.resize_rgb_mask(render_track_rgb, render_track_mask, render_h, render_w)
            if not self.use_tracker:
                render_dyn_rgb_final, render_dyn_mask_final = render_dyn_rgb, render_dyn_mask

        info_dict = {
            "temporal_closest_rgb": render_dyn_rgb,
            "temporal_closest_mask": render_dyn_mask,
            "temporal_track_rgb": render_track_rgb,
            "temporal_track_mask": render_track_mask,
        }
        if fuse_static:
            st = (lambda k: torch.stack([c[k] for c in combs], 0)) if n_b > 1 else (lambda k: combs[0][k][None])
            info_dict["combined_rgb"] = out_comb if out_comb is not None else st(0)
            info_dict["combined_rgb_static"] = st(1)
            info_dict["combined_rgb_dyn"] = st(2)
        return render_dyn_rgb_final, render_dyn_mask_final, info_dict

    _RNG_SLOTS = 64

    def splat_rng_state(self, dev):
        """Device-resident {seed, draw number} of the splat kernel's own noise for the CURRENT stream of ``dev``: one
        state per stream, so that views in flight on different streams neither share draws nor race on the counter.
        All states of a device live in one block of ``_RNG_SLOTS`` rows made at the first call for that device (a single
        host-to-device copy; nothing is allocated later, e.g. under a graph capture on a stream first seen there);
        a process that uses more streams than rows wraps around (two streams then share a counter: their draws stay
        valid normal fields, only no longer distinct per stream)."""
        pools = self.__dict__.setdefault("_splat_rng", {})
        pool = pools.get(dev)
        if pool is None:
            seeds = [((torch.initial_seed() + 0x9E3779B97F4A7C15 * i) & 0x7FFFFFFFFFFFFFFF, 0) for i in range(self._RNG_SLOTS)]
            pool = pools[dev] = {"block": torch.tensor(seeds, dtype=torch.int64, device=dev), "slots": {}}
        key = torch.cuda.current_stream(dev).cuda_stream
        slot = pool["slots"].get(key)
        if slot is None:
            if len(pool["slots"]) >= 4 * self._RNG_SLOTS:  # (stream handles come and go: forget the oldest mapping)
                pool["slots"].pop(next(iter(pool["slots"])))
            slot = pool["slots"][key] = pool.setdefault("next", 0) % self._RNG_SLOTS
            pool["next"] = pool["next"] + 1
        return pool["block"][slot]

    def _zeros_like(self, t):
        """zero images of the no-tracker outputs without filling 33 MB per view at 1080p: ONE zero element expanded to
        the shape (stride 0).  Reads behave like a zeros tensor; an in-place write raises (torch refuses to write
        through overlapping memory), so a caller cannot corrupt what later views return -- ``.clone()`` it to edit, or set
        ``PGDVS_MATERIALIZE_ZEROS=1`` for fresh writable zero tensors as upstream returns them (33 MB filled per 1080p view)."""
        import os

        if os.environ.get("PGDVS_MATERIALIZE_ZEROS") == "1":  # upstream's behaviour to the letter: a fresh, writable tensor per view
            return torch.zeros_like(t)
        cache = self.__dict__.setdefault("_zero_cache", {})
        key = (t.dtype, t.device)
        z = cache.get(key)
        if z is None:
            z = cache[key] = torch.zeros(1, dtype=t.dtype, device=t.device)
        return z.expand(t.shape)

    def resize_rgb_mask(self, rgb, mask, render_h, render_w):
        # :259-270 -- only taken when render_stride != 1; torch resampling (plumbing, GPU)
        rgb = torch.nn.functional.interpolate(rgb, size=(render_h, render_w), mode="bicubic", align_corners=True, antialias=True)
        mask = torch.nn.functional.interpolate(mask, size=(render_h, render_w), mode="nearest")
        return rgb, mask
