"""Mirror of ``pgdvs.renderers.pgdvs_renderer_dyn_track.PGDVSDynamicTrackRenderer``
(pgdvs/renderers/pgdvs_renderer_dyn_track.py:26-764), row A17 of SURVEY.md 8a.

The point trackers (TAPIR / CoTracker) are third-party pretrained networks outside this
build.  Everything around them is here: ``prepare_data`` (frame window assembly),
``run_track`` / ``run_track_func`` (query construction + chunked tracker calls, for a
plugged-in tracker with the reference's interface), ``compute_pcl_for_tgt`` (HIP: validity,
frame-pair selection, sampling, unprojection, time interpolation, track-to-base and
statistical filters, concatenation -- all with device-side counts) and ``render_with_track``.

Without a tracker module the tracks are read from the data dict:
``data["track_tracks"][i_b]`` [#pt, N, 2] (col,row), ``data["track_visibles"][i_b]`` [#pt, N]
(and optionally ``data["track_query_pts"][i_b]`` [#pt, 3]), with the N frames ordered as
``prepare_data`` orders them: [fwd2tgt..., temporally-closest..., bwd2tgt...].
"""
import numpy as np
import torch

from .. import ops
from .pgdvs_renderer_dyn import PGDVSDynamicRenderer

KIND_CLOSEST, KIND_TRACK = 1, 2


class PGDVSDynamicTrackRenderer(PGDVSDynamicRenderer):
    # -- :27-96 -------------------------------------------------------------------
    def render_with_track(self, data, render_cfg, base_pcl_info, for_debug=False, disable_tqdm=False, cams_tgt=None):
        device = data["rgb_src_temporal_track_fwd2tgt"].device
        n_b, n_views_one_side, orig_h, orig_w, _ = data["rgb_src_temporal_track_fwd2tgt"].shape
        n_views = n_views_one_side * 2 + 2  # last 2 is for the two temporally-closest source views
        if cams_tgt is None:
            cams_tgt = ops.cam_prep(data["flat_cam_tgt"])

        track_rgbs, track_masks = [], []
        for i_b in range(n_b):
            data_for_track = self.prepare_data(i_b, data, n_views, device)
            if self.tracker is not None:
                dyn_mask_for_track = data_for_track["dyn_masks_for_track"][data_for_track["idx_real_track"], ...]
                if not bool(torch.sum(dyn_mask_for_track) > 0):  # :49 (host decision upstream as well)
                    tracks = None
                else:
                    _, tracks, track_visibles = self.run_track(data_for_track, for_debug=for_debug, disable_tqdm=disable_tqdm)
            else:
                if "track_tracks" not in data:
                    raise KeyError(
                        "dyn_render_track_temporal='no_tgt' without a tracker module: supply data['track_tracks'] / "
                        "data['track_visibles'] (see module docstring)")
                tracks, track_visibles = data["track_tracks"][i_b], data["track_visibles"][i_b]
                if tracks is not None and tracks.shape[0] == 0:
                    tracks = None

            if tracks is None:
                track_rgbs.append(torch.zeros((3, orig_h, orig_w), dtype=torch.float32, device=device))
                track_masks.append(torch.zeros((1, orig_h, orig_w), dtype=torch.float32, device=device))
                continue
            cur_base = {k: base_pcl_info[k][i_b] for k in ("pcl", "pcl_rgbs", "pcl_nn_dist_thres")}
            cur_base["n_pts"] = base_pcl_info["n_pts"][i_b] if "n_pts" in base_pcl_info else None
            pcl, rgbs, n_pts = self.compute_pcl_for_tgt(
                data_for_track=data_for_track, query_pts=None, tracks=tracks, track_visibles=track_visibles,
                render_cfg=render_cfg, base_pcl_info=cur_base, device=device, return_count=True)
            rgb, mask = self.render_dyn_pcl(pcl=pcl, rgbs=rgbs, n_pts=n_pts, cam_tgt=cams_tgt[i_b], H=orig_h, W=orig_w,
                                            render_cfg=render_cfg)
            track_rgbs.append(rgb)
            track_masks.append(mask[None])
        return torch.stack(track_rgbs, 0), torch.stack(track_masks, 0)  # [B,3,H,W], [B,1,H,W]

    # -- :98-396 ------------------------------------------------------------------
    def compute_pcl_for_tgt(self, *, data_for_track, query_pts, tracks, track_visibles, render_cfg, base_pcl_info,
                            device=None, for_debug=False, return_count=False):
        """Point cloud of the tracker window at the target time.

        ``base_pcl_info``: {"pcl", "pcl_rgbs", "pcl_nn_dist_thres"[, "n_pts"]} of the
        temporally-closest rendering (None entries = no base cloud, :146-152 of the dyn
        renderer).  ``n_pts`` (device int32) lets ``pcl`` be a capacity-sized buffer.
        Returns (pcl, rgbs) trimmed to the exact size like upstream (one host read), or the
        capacity-sized buffers plus the device count with ``return_count=True``.
        ``query_pts`` is unused (as upstream)."""
        K = int(render_cfg.dyn_pcl_outlier_knn)
        dft = data_for_track
        valid, pcl_all, rgb_all = ops.track_points(
            tracks, track_visibles, dft["frame_kind"], dft["time_for_track_raw"], dft["time_tgt_raw"],
            dft["rgbs_for_track"][: dft["n_actual_frames"]], dft["depths_for_track"][..., 0], dft["cams_for_track"])
        idx, cnt = ops.compact_u8(valid)
        pcl = ops.gather_rows(pcl_all, idx, cnt)
        rgb = ops.gather_rows(rgb_all, idx, cnt)

        base_pcl = base_pcl_info["pcl"]
        has_base = base_pcl is not None and base_pcl.shape[0] > 0
        base_cnt = None
        if has_base:
            base_cnt = base_pcl_info.get("n_pts", None)
            if base_cnt is None:
                base_cnt = torch.full((1,), base_pcl.shape[0], dtype=torch.int32, device=base_pcl.device)
            base_thres = base_pcl_info["pcl_nn_dist_thres"].reshape(1).float()
            # :295-328 remove points that are too far away from the base cloud
            avg = ops.knn_cross_mean_dist(pcl, cnt, base_pcl, base_cnt, K + 1)
            flag = ops.threshold_flags(avg, cnt, base_thres, float(render_cfg.dyn_pcl_track_track2base_thres_mult),
                                       None, base_cnt)
            idx, cnt = ops.compact_u8(flag)
            pcl, rgb = ops.gather_rows(pcl, idx, cnt), ops.gather_rows(rgb, idx, cnt)

        # :340-380 statistical outlier removal; the base cloud's threshold when there is one
        avg = ops.knn_mean_dist(pcl, cnt, K)
        own_thres, flag = ops.outlier_flags(avg, cnt, render_cfg.dyn_pcl_outlier_std_thres, True)
        if has_base:
            flag = ops.threshold_flags(avg, cnt, base_thres, 1.0, own_thres, base_cnt)
        idx, cnt = ops.compact_u8(flag)
        pcl, rgb = ops.gather_rows(pcl, idx, cnt), ops.gather_rows(rgb, idx, cnt)

        # :390-394 the base cloud is appended to a non-empty track cloud
        if has_base:
            pcl, n_out = ops.concat_rows(pcl, cnt, base_pcl, base_cnt, require_a=True)
            rgb, _ = ops.concat_rows(rgb, cnt, base_pcl_info["pcl_rgbs"], base_cnt, require_a=True)
        else:
            n_out = cnt
        if return_count:
            return pcl, rgb, n_out
        n = int(n_out.item())
        return pcl[:n], rgb[:n]

    # -- :398-558 (only with a plugged-in tracker module) ------------------------------
    def run_track(self, data_for_track, for_debug=False, disable_tqdm=False):
        if self.tracker is None:
            raise RuntimeError("run_track needs a tracker module (cfg.tracker); none is configured")
        if not getattr(self.tracker, "separate_fwd_bwd", False):  # TAPNet-style: one pass over the window (:411-418)
            return self.run_track_func(data_for_track, track_k="idx_real_track", track_bwd=False, disable_tqdm=disable_tqdm)
        # CoTracker-style: forward and backward halves separately (:419-459)
        outs = []
        for key, bwd in (("idx_real_track_fwd", False), ("idx_real_track_bwd", True)):
            if len(data_for_track[key]) > 0 and bool(torch.sum(data_for_track["dyn_masks_for_track"][data_for_track[key], ...]) > 0):
                outs.append(self.run_track_func(data_for_track, track_k=key, track_bwd=bwd, disable_tqdm=disable_tqdm))
        return tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(3))

    def run_track_func(self, data_for_track, track_k="idx_real_track", track_bwd=False, for_debug=False, disable_tqdm=False):
        query_pts = []
        for idx in data_for_track[track_k]:
            rows, cols = torch.nonzero(data_for_track["dyn_masks_for_track"][idx, ..., 0] > 0.0, as_tuple=True)
            query_pts.append(torch.stack((torch.ones_like(rows) * idx, rows, cols), dim=1).float())
        query_pts = torch.cat(query_pts, dim=0)  # [#pt, 3] (time, row, col)
        n_actual_pts = query_pts.shape[0]
        chunk = self.track_chunk_size
        n_final_pts = int(np.ceil(n_actual_pts / chunk) * chunk)  # fixed shapes for the tracker (:492-498)
        query_pts = torch.cat((query_pts, query_pts[: n_final_pts - n_actual_pts, ...]), dim=0)
        n_actual_frames = data_for_track["n_actual_frames"]
        rgbs = data_for_track["rgbs_for_track"]
        n_frames = rgbs.shape[0]
        if track_bwd:
            rgbs = torch.flip(rgbs[:n_actual_frames, ...], dims=[0])
            if n_actual_frames != n_frames:
                rgbs = torch.cat((rgbs, rgbs[-(n_frames - n_actual_frames):, ...]), dim=0)
            query_pts[:, 0] = n_actual_frames - 1 - query_pts[:, 0]
        tracks, visibles = [], []
        for start in range(0, n_final_pts, chunk):
            tmp_tracks, tmp_visibles = self.tracker(frames=rgbs, query_points=query_pts[start:start + chunk, :])
            if track_bwd:
                tmp_tracks = torch.flip(tmp_tracks[:, :n_actual_frames, :], dims=[1])
                tmp_visibles = torch.flip(tmp_visibles[:, :n_actual_frames], dims=[1])
            tracks.append(tmp_tracks)
            visibles.append(tmp_visibles)
        tracks = torch.cat(tracks, dim=0)[:n_actual_pts, :n_actual_frames, :]  # (col,row)
        visibles = torch.cat(visibles, dim=0)[:n_actual_pts, :n_actual_frames]
        if track_bwd:
            query_pts[:, 0] = n_actual_frames - 1 - query_pts[:, 0]
        return query_pts[:n_actual_pts, ...], tracks, visibles

    # -- :599-764 -----------------------------------------------------------------
    def prepare_data(self, i_b, data, n_views, device):
        parts = {k: [] for k in ("rgb", "dyn_mask", "depth", "flat_cam", "time")}
        idx_temporal_closest, idx_real_track, idx_fwd, idx_bwd = [], [], [], []
        n_actual_frames = 0
        for suffix, nkey in (("_track_fwd2tgt", "n_actual_temporal_track_fwd2tgt"), ("", "n_actual_temporal"),
                             ("_track_bwd2tgt", "n_actual_temporal_track_bwd2tgt")):
            n = int(data[nkey][i_b, 0])
            if n <= 0:
                assert suffix != "", "no temporally-closest source view"
                continue
            for k in parts:
                parts[k].append(data[f"{k}_src_temporal{suffix}"][i_b, :n])
            ids = list(range(n_actual_frames, n_actual_frames + n))
            if suffix == "":
                idx_temporal_closest = ids
            else:
                idx_real_track.extend(ids)
                (idx_fwd if suffix == "_track_fwd2tgt" else idx_bwd).extend(ids)
            n_actual_frames += n

        rgbs = torch.cat(parts["rgb"], dim=0)  # [N,H,W,3]
        dyn_masks = torch.cat(parts["dyn_mask"], dim=0)  # [N,H,W,1]
        depths = torch.cat(parts["depth"], dim=0)  # [N,H,W,1]
        flat_cams = torch.cat(parts["flat_cam"], dim=0)  # [N,34]
        times_raw = torch.cat(parts["time"], dim=0).float()
        min_time = torch.min(times_raw)
        frame_kind = [KIND_CLOSEST if i in idx_temporal_closest else KIND_TRACK for i in range(n_actual_frames)]

        # fixed-shape padding for the tracker networks (:723-731)
        n_rep = int(np.ceil(n_views / n_actual_frames))
        ret = {
            "n_actual_frames": n_actual_frames,
            "rgbs_for_track": rgbs.repeat(n_rep, 1, 1, 1)[:n_views, ...] if self.tracker is not None else rgbs,
            "dyn_masks_for_track": dyn_masks.repeat(n_rep, 1, 1, 1)[:n_views, ...] if self.tracker is not None else dyn_masks,
            "depths_for_track": depths,
            "flat_cams_for_track": flat_cams,
            "time_for_track": times_raw - min_time,  # normalised to start from 0 (:718-719)
            "time_tgt": data["time_tgt"][i_b, :].float() - min_time,
            "idx_temporal_closest": idx_temporal_closest,
            "idx_real_track": idx_real_track,
            "idx_real_track_fwd": idx_fwd,
            "idx_real_track_bwd": idx_bwd,
            # inputs of the HIP kernel (it applies the same shift itself)
            "frame_kind": frame_kind,
            "time_for_track_raw": times_raw.contiguous(),
            "time_tgt_raw": data["time_tgt"][i_b, :1].float().contiguous(),
            "cams_for_track": ops.cam_prep(flat_cams),
        }
        ret["time_real_track"] = ret["time_for_track"][idx_real_track]
        return ret
