"""Mirror of ``pgdvs.models.gnt.renderer.BaseRenderer`` (pgdvs/models/gnt/renderer.py:21-485):
the GNT static-scene renderer -- feature extraction on all source views, then per chunk of
rays: sampling + epipolar gathering (HIP, A13), view/ray transformer aggregation (A14) and
the per-ray reductions (A15).  Same constructor (``model_cfg`` with a ``_target_``) and the
same ``forward(ray_batch=..., chunk_size=..., ...)`` contract and output dictionary."""
from collections import OrderedDict

import torch

from ... import ops
from ...instantiate import instantiate
from .projector import Projector


class BaseRenderer(torch.nn.Module):
    def __init__(self, *, model_cfg=None):
        super().__init__()
        self.projector = Projector()
        self.model = None
        # measurement hook (bench.py): a dict set here is filled by `forward` with (start, end) event pairs per
        # stage -- "features" (ResUNet), "gather" (A13), "transformer" (A14 + the per-ray reductions); None = no
        # events recorded
        self.stage_events = None
        # `chunk_size` bounds memory upstream (renderer.py:414-485: one chunk of rays at a time) and changes no result -- rays
        # are independent.  Here consecutive chunks are merged into execution chunks of up to this many rays when the
        # device has the memory for them (the fused kernels are persistent workgroups: 1024 rays x 256 samples are 8 rounds
        # of tiles per launch with a tail of one, 4096 rays 32 rounds; the 288 x 550 view takes 5 % less GPU time).
        # 0 = execute chunk by chunk as given.
        self.merge_chunks_up_to = 4096
        if model_cfg is not None and (model_cfg.get("_target_", None) if hasattr(model_cfg, "get") else None):
            self.model = instantiate(model_cfg)
        elif model_cfg is not None:
            from .model import GNTModel

            kw = {k: v for k, v in dict(model_cfg).items() if k != "_target_"}
            self.model = GNTModel(**kw)

    def forward(self, *, ray_batch, chunk_size, inv_uniform=False, n_coarse_samples_per_ray, n_fine_samples_per_ray=0,
                flag_deterministic=False, use_dyn_mask=False, render_stride=1, ret_view_entropy=False, ret_view_std=False,
                debug_epipolar=False, disable_tqdm=False):
        if self.model is None:
            raise RuntimeError("BaseRenderer was built without a model_cfg: no GNT network to run")
        if not flag_deterministic:
            raise NotImplementedError("stochastic ray sampling is a training-time feature; PGDVS renders with flag_deterministic=True")
        src_rgbs = ray_batch["src_rgbs"]  # [B,V,H,W,3]
        B, V, H, W, _ = src_rgbs.shape
        n_rays = ray_batch["ray_o"].shape[0]
        ev = self.stage_events

        def timed(stage, fn):
            if ev is None:
                return fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()  # on the stream that is current where the stage runs
            out = fn()
            b.record()
            ev.setdefault(stage, []).append((a, b))
            return out

        raw_feats = timed("features", lambda: self.model.feature_net(src_rgbs.permute(0, 1, 4, 2, 3).reshape(B * V, 3, H, W)))  # (coarse, fine)
        to_cl = lambda f: f.permute(0, 2, 3, 1).contiguous().reshape((B, V) + tuple(f.shape[2:]) + (f.shape[1],))  # noqa: E731
        feats_cl = to_cl(raw_feats[0])
        n_fine = int(n_fine_samples_per_ray)
        feats_fine_cl = (feats_cl if raw_feats[1] is raw_feats[0] else to_cl(raw_feats[1])) if n_fine > 0 else None
        cams_src = ops.cam_prep(ray_batch["src_cameras"])  # [B,V,80]
        cams_tgt = ops.cam_prep(ray_batch["camera"])  # [B,80]
        per_ray_range = bool(ray_batch["depth_range_per_ray"])
        inv_masks = ray_batch["src_invalid_masks"][..., 0] if use_dyn_mask else None  # [B,V,H,W]
        rays_per_view = n_rays // B
        if chunk_size < 0:
            chunk_size = n_rays
        merge = int(self.merge_chunks_up_to or 0)
        if merge >= 2 * chunk_size and src_rgbs.is_cuda:
            # gathered block + embedded features + transformer temporaries of an execution chunk: ~900 bytes per (ray, sample,
            # view) with room to spare; merged only while that is a tenth of the free memory
            m = merge // chunk_size
            n_all = int(n_coarse_samples_per_ray) + int(n_fine_samples_per_ray)
            # (the factor is worked out ONCE per (device, chunk size, samples, views) and kept: `mem_get_info` is a driver call,
            # and what it reports moves with the caching allocator's state -- asked per forward, the launch structure of a
            # view depended on it.  Memory the allocator holds but has free counts as free.)
            key = (src_rgbs.device.index, int(chunk_size), n_all, int(V), merge)
            cache = self.__dict__.setdefault("_merge_factor", {})
            if key not in cache:
                dev_free = torch.cuda.mem_get_info(src_rgbs.device)[0]
                pooled = torch.cuda.memory_reserved(src_rgbs.device) - torch.cuda.memory_allocated(src_rgbs.device)
                free = dev_free + max(int(pooled), 0)
                while m > 1 and m * chunk_size * n_all * V * 900 > free // 10:
                    m -= 1
                cache[key] = m
            chunk_size *= cache[key]
        outs, outs_fine = OrderedDict(), OrderedDict()
        # a chunk may straddle batch items (true batching, renderer.py:414-485): one job per (chunk, batch item)
        jobs = []
        for ci, c0 in enumerate(range(0, n_rays, chunk_size)):
            c1 = min(c0 + chunk_size, n_rays)
            for b in range(c0 // rays_per_view, (c1 - 1) // rays_per_view + 1):
                jobs.append((ci, b, max(c0, b * rays_per_view), min(c1, (b + 1) * rays_per_view)))
        drange = lambda b, lo, hi: ray_batch["depth_range"][lo:hi] if per_ray_range else ray_batch["depth_range"][b:b + 1]  # noqa: E731

        def gather(job):  # the coarse pass's A13 of one job
            _, b, lo, hi = job
            return timed("gather", lambda: ops.gnt_gather(
                ray_batch["ray_o"][lo:hi], ray_batch["ray_d"][lo:hi], drange(b, lo, hi), n_coarse_samples_per_ray,
                inv_uniform, cams_tgt[b], cams_src[b], src_rgbs[b], feats_cl[b], None if inv_masks is None else inv_masks[b]))

        pieces, cur = [], -1

        def flush():
            for dst, which in ((outs, 0), (outs_fine, 1)):
                if pieces[0][which] is None:
                    continue
                for k in pieces[0][which]:
                    dst.setdefault(k, []).append(
                        pieces[0][which][k] if len(pieces) == 1 else torch.cat([p[which][k] for p in pieces], 0))

        for ci, b, lo, hi in jobs:
            g = gather((ci, b, lo, hi))
            if ci != cur and pieces:
                flush()
                pieces = []
            cur = ci
            pieces.append(timed("transformer", lambda: self._render_rays(  # noqa: B023 -- called at once
                g=g, ray_o=ray_batch["ray_o"][lo:hi], ray_d=ray_batch["ray_d"][lo:hi], depth_range=drange(b, lo, hi),
                cam_tgt=cams_tgt[b], cams_src=cams_src[b], src_rgbs=src_rgbs[b],
                inv_masks=None if inv_masks is None else inv_masks[b],
                inv_uniform=inv_uniform, ret_view_entropy=ret_view_entropy, ret_view_std=ret_view_std,
                n_fine=n_fine, feats_fine_cl=None if feats_fine_cl is None else feats_fine_cl[b])))
            del g
        if pieces:
            flush()
        rh = (ray_batch["raw_h"] + render_stride - 1) // render_stride
        rw = (ray_batch["raw_w"] + render_stride - 1) // render_stride
        cat = lambda d: OrderedDict((k, torch.cat(v, dim=0).reshape((B, rh, rw, -1))) for k, v in d.items())  # noqa: E731
        return OrderedDict([("outputs_coarse", cat(outs)), ("outputs_fine", cat(outs_fine) if n_fine > 0 else None)])

    def _render_rays(self, *, g, ray_o, ray_d, depth_range, cam_tgt, cams_src, src_rgbs, inv_masks,
                     inv_uniform, ret_view_entropy, ret_view_std, n_fine=0, feats_fine_cl=None):
        """render_rays (:207-412) for rays of one batch item, `g` = the coarse pass's gathered block
        -> (coarse outputs, fine outputs or None)."""
        from .ray_sampler import sample_fine_z

        V = src_rgbs.shape[0]

        def one_pass(net, g):
            out, extras = net(g["rgb_feat"], g["ray_diff"], g["mask"], g["pts"], ray_d,
                              ret_view_entropy=ret_view_entropy, ret_view_std=ret_view_std)
            rgb, weights = out[:, 0:3], out[:, 3:]
            ret = {
                "rgb": rgb, "weights": weights, "depth": torch.sum(weights * g["z_vals"], dim=-1),
                "inbound_cnt": torch.sum(weights * g["mask_inbound"][..., 0].sum(dim=2) / V, dim=1),
                "dyn_cnt": torch.sum(weights * g["mask_invalid"][..., 0].sum(dim=2) / V, dim=1),
            }
            for k in ("view_entropy", "view_std", "view_std_normalized"):
                if k in extras:
                    ret[k] = torch.sum(weights[..., None] * extras[k], dim=1)
            return ret

        coarse = one_pass(self.model.net_coarse, g)
        if n_fine <= 0:
            return coarse, None
        # importance re-sampling from the coarse weights, second pass on the fine features (:313-412)
        z_all = sample_fine_z(inv_uniform, n_fine, True, coarse["weights"].detach().clone(), g["z_vals"])
        gf = ops.gnt_gather(ray_o, ray_d, depth_range, z_all.shape[1], inv_uniform, cam_tgt, cams_src, src_rgbs,
                            feats_fine_cl, inv_masks, z_samples=z_all)
        net = self.model.net_coarse if self.model.single_net else self.model.net_fine
        return coarse, one_pass(net, gf)
