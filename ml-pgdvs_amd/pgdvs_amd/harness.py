"""Caller-side harness that makes end-to-end numbers comparable with the reference's evaluator
(SURVEY.md 8f-1): ``eval_step`` reproduces ``PGDVSEvaluator.eval_step``
(pgdvs/engines/evaluator_pgdvs.py:26-188) around any renderer with the plugin contract -- to-device,
``forward`` under no_grad, clamp -> NaN to 0 -> quantise, ground truth resized to the render size,
the evaluator's three masked PSNRs (``obtain_quantitative_nvidia`` :190-283 with
``calculate_psnr`` pgdvs/utils/training.py:281-313) and ONE packed reduce of the metric sums to rank 0
(the reference issues one ``torch.distributed.reduce`` per key, :183-186).  LPIPS and SSIM need
third-party networks / skimage and are out of scope: their keys are not produced.  A few elementwise
torch ops on final images; nothing here is on the rendering hot path."""
import math
import time
from collections import OrderedDict

import torch

from . import dist as pdist


def quantize_like_evaluator(x: torch.Tensor) -> torch.Tensor:
    """clamp(0,1) -> NaN to 0 -> (x*255).byte().float()/255 (evaluator_pgdvs.py:52-77)."""
    x = torch.nan_to_num(x.clamp(0.0, 1.0), nan=0.0)
    return (x * 255).byte().float() / 255.0


def masked_psnr(img1: torch.Tensor, img2: torch.Tensor, mask: torch.Tensor) -> float:
    """calculate_psnr (training.py:281-313): float64, mse normalised by sum(mask)+1e-8 (the mask
    broadcasts over channels exactly as given), and the reference's quirk of returning 0 when
    the images are identical."""
    assert img1.ndim == 3 and img2.ndim == 3
    a, b, m = img1.double(), img2.double(), mask.double()
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and float(b.min()) >= 0 and float(b.max()) <= 1
    mse = float((((a - b) ** 2) * m).sum() / (m.sum() + 1e-8))
    if mse == 0:
        return 0
    return 10 * math.log10(1.0 / mse)


def to_device(batch: dict, device) -> dict:
    """``_to_gpu_func`` (pgdvs/engines/abstract.py:153-157): tensors move, everything else passes through"""
    return {k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}


METRIC_KEYS = ("psnr_full_combined", "psnr_dyn_combined", "psnr_static_combined")

# measurement hook (bench.py): a dict set here accumulates the host wall time of eval_step's stages in seconds
# ("to_device", "forward" = enqueue of the renderer, "metric_enqueue", "sync_read" = the step's one wait for the GPU,
# "post"); None = no timing
STAGE_SECONDS = None


@torch.no_grad()
def eval_step(model, data: dict, render_cfg, *, device=None, disable_tqdm=True, return_images=False):
    """One evaluator step on a batch of target views.  ``data`` is the reference's data dict (row A0) plus
    ``rgb_tgt[B,H,W,3]`` and ``eval_mask[B,H,W,3]`` (1 = dynamic region).  Returns the reference's
    ``metric_dict`` restricted to the in-scope keys: ``eval/count`` (int64) and the per-key SUMS over the
    batch (float32), reduced to rank 0 when a process group is up (device tensors then, as upstream; in a single process
    HOST tensors on both the fused GPU path and the torch path, so that a caller who accumulates them over steps never
    mixes devices).  With ``return_images`` also the quantised prediction / ground truth and the
    per-view values."""
    device = device if device is not None else next(iter(v for v in data.values() if isinstance(v, torch.Tensor))).device
    stages, t_prev = STAGE_SECONDS, time.perf_counter()

    def lap(name):
        nonlocal t_prev
        if stages is not None:
            now = time.perf_counter()
            stages[name] = stages.get(name, 0.0) + (now - t_prev)
            t_prev = now

    data_gpu = to_device(data, device)
    lap("to_device")
    if model.training:
        model.eval()
    n_batch = data["rgb_src_temporal"].shape[0]
    ret = model.forward(data_gpu, render_cfg=render_cfg, disable_tqdm=disable_tqdm, for_debug=False)
    lap("forward")
    from . import ops

    def check_status(host_counts=None, host_status=None):
        # device-side status words of the geometry path (the step synchronises for its metrics anyway): a static cloud whose
        # aggregation reported an error (count -1), filled its buffer (rows may have been dropped: the aggregation clamps
        # at its capacity) or outgrew the rasteriser's row bound would otherwise show up as a silently blank or truncated
        # static image.  host_counts / host_status: the words as already read back with the metric sums.
        cnts = ret.get("st_pcl_rgb_count", data_gpu.get("st_pcl_rgb_count", None))
        if isinstance(cnts, torch.Tensor):
            cloud = ret.get("st_pcl_rgb", None)
            values = host_counts if host_counts is not None else [int(c.item()) for c in cnts.reshape(-1)]
            for n in values:
                if n < 0:
                    raise ops.PgdvsHipError(f"st_pcl_rgb_count: device-side error flag set (count {n}); the output is not valid")
                limited = cloud is not None and "_st_pcl_video" in data_gpu and cloud.shape[1] < data_gpu["_st_pcl_video"]["depths"].numel()
                if limited and n >= cloud.shape[1]:
                    raise ops.PgdvsHipError(f"the aggregated static cloud filled its buffer of {cloud.shape[1]} rows (capacity-limited): "
                                            "rows may have been dropped -- pass a larger capacity")
        if host_status is not None:
            ops.check_raster_status(torch.tensor(host_status, dtype=torch.int32))
        else:
            ops.check_raster_status(ret.get("geo_static_raster_status", None))

    comb = ret["combined_rgb"]
    if (comb.is_cuda and comb.dtype == torch.float32 and tuple(comb.shape[2:]) == tuple(data_gpu["rgb_tgt"].shape[1:3])
            and data_gpu["rgb_tgt"].dtype == torch.float32 and data_gpu["eval_mask"].dtype == torch.float32):
        # GPU, render size == ground-truth size (render_stride 1): quantisation and the three masked sums of a view in ONE
        # pass (csrc/eval.hip), one host read for the whole batch
        # (the device-side status words of the geometry path ride along in the same block: ONE host read per batch)
        cnts = ret.get("st_pcl_rgb_count", data_gpu.get("st_pcl_rgb_count", None))
        cnts = cnts.reshape(-1) if isinstance(cnts, torch.Tensor) and cnts.is_cuda and cnts.dtype == torch.int64 else None
        stat = ret.get("geo_static_raster_status", None)
        stat = stat.reshape(-1) if isinstance(stat, torch.Tensor) and stat.is_cuda and stat.dtype == torch.int32 else None
        res = [ops.eval_psnr_sums(comb[i_b], data_gpu["rgb_tgt"][i_b], data_gpu["eval_mask"][i_b], want_images=return_images,
                                  count_dev=cnts[i_b:i_b + 1] if (cnts is not None and i_b < cnts.numel()) else None,
                                  status_dev=stat[i_b:i_b + 1] if (stat is not None and i_b < stat.numel()) else None)
               for i_b in range(n_batch)]
        lap("metric_enqueue")
        sums = ops.read_back_rows([r_[0] for r_ in res])  # (the step's synchronisation)
        lap("sync_read")
        check_status(host_counts=[int(s_[6]) for s_ in sums] if cnts is not None else None,
                     host_status=[int(s_[7]) for s_ in sums] if stat is not None else None)
        per_view = {k: [] for k in METRIC_KEYS}
        for s_ in sums:
            for j, k in enumerate(METRIC_KEYS):
                mse = s_[j] / (s_[3 + j] + 1e-8)
                per_view[k].append(0 if mse == 0 else 10 * math.log10(1.0 / mse))
        # (a single process keeps the packed sums -- and so the metric tensors -- on the host: same dtypes and values, no
        # upload and no one-element kernels per step, and the caller's `.item()` costs nothing; ranks that reduce over
        # RCCL need them on the device, like upstream)
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if not multi:
            metric = {"eval/count": torch.tensor([n_batch], dtype=torch.int64)}
            for k in METRIC_KEYS:
                metric[f"eval/{k}"] = torch.tensor(per_view[k], dtype=torch.float32).sum()
        else:
            packed = torch.tensor([float(n_batch)] + [float(torch.tensor(per_view[k], dtype=torch.float32).sum()) for k in METRIC_KEYS],
                                  dtype=torch.float64, device=comb.device)
            packed = pdist.reduce_metrics(packed, dst=0)
            metric = {"eval/count": packed[:1].round().to(torch.int64)}
            for j, k in enumerate(METRIC_KEYS):
                metric[f"eval/{k}"] = packed[1 + j].to(torch.float32)
        lap("post")
        if return_images:
            return metric, {"pred": torch.stack([r_[1] for r_ in res]), "gt": torch.stack([r_[2] for r_ in res]),
                            "eval_mask": data_gpu["eval_mask"].permute(0, 3, 1, 2), "per_view": per_view, "ret": ret}
        return metric
    check_status()
    pred = OrderedDict({"combined": ret["combined_rgb"].clamp(0.0, 1.0)})
    for k in pred:
        if torch.any(torch.isnan(pred[k])):
            pred[k] = torch.nan_to_num(pred[k], nan=0.0)
    rgb_gt = data_gpu["rgb_tgt"].permute(0, 3, 1, 2).clamp(0.0, 1.0)
    eval_mask = data_gpu["eval_mask"].permute(0, 3, 1, 2)
    # quantise first, as if the images had been written to disk and read back (:70-77)
    rgb_gt = (rgb_gt * 255).byte().float() / 255.0
    for k in pred:
        pred[k] = (pred[k] * 255).byte().float() / 255.0
    _, _, rh, rw = pred["combined"].shape
    if rgb_gt.shape[2] != rh or rgb_gt.shape[3] != rw:  # render_stride != 1 (:80-92)
        rgb_gt = torch.nn.functional.interpolate(rgb_gt, size=(rh, rw), mode="bicubic", antialias=True, align_corners=True)
        eval_mask = torch.nn.functional.interpolate(eval_mask, size=(rh, rw), mode="nearest")
        eval_mask = (eval_mask > 0).float()
    per_view = {k: [] for k in METRIC_KEYS}
    for i_b in range(n_batch):
        p, g = pred["combined"][i_b].to(rgb_gt.device), rgb_gt[i_b]
        m_dyn = eval_mask[i_b]
        # calculate_psnr asserts on [0,1] inputs; a bicubically resized ground truth can overshoot, as upstream
        per_view["psnr_full_combined"].append(masked_psnr(g, p, torch.ones_like(g)))
        per_view["psnr_dyn_combined"].append(masked_psnr(g, p, m_dyn))
        per_view["psnr_static_combined"].append(masked_psnr(g, p, 1.0 - m_dyn))
    # one packed reduce instead of one collective per key: [count, sums...] in float64 on the device
    packed = torch.tensor([float(n_batch)] + [float(torch.tensor(per_view[k], dtype=torch.float32).sum()) for k in METRIC_KEYS],
                          dtype=torch.float64, device=rgb_gt.device)
    packed = pdist.reduce_metrics(packed, dst=0)
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    if not multi:
        packed = packed.cpu()  # one process: host tensors on BOTH paths, whatever the inputs' dtypes and sizes were
    metric = {"eval/count": packed[:1].round().to(torch.int64)}
    for j, k in enumerate(METRIC_KEYS):
        metric[f"eval/{k}"] = packed[1 + j].to(torch.float32)
    if return_images:
        return metric, {"pred": pred["combined"], "gt": rgb_gt, "eval_mask": eval_mask, "per_view": per_view, "ret": ret}
    return metric
