"""Caller-side helpers that make end-to-end numbers comparable with the reference's evaluator
(SURVEY.md 8f-1): the quantisation of predictions before metrics
(pgdvs/engines/evaluator_pgdvs.py:52-77) and the masked PSNR
(pgdvs/utils/training.py:281-313).  Metric plumbing only -- a few elementwise torch ops on
the final images; nothing here is on the rendering hot path."""
import math

import torch


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
