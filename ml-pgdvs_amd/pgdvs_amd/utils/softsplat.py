"""Drop-in for ``pgdvs.utils.softsplat`` (pgdvs/utils/softsplat.py:280-333).

Same signature, modes and assertions; the cupy/NVRTC kernel ``softsplat_out``
(softsplat.py:352-402) is replaced by the gfx950 kernel behind
``pgdvs_softsplat_fwd`` with the exp / premultiply / normalise steps fused in.
Forward only (the reference's engines run under ``torch.no_grad()``).
"""
import torch

from .. import ops


def softsplat(tenIn: torch.Tensor, tenFlow: torch.Tensor, tenMetric: torch.Tensor, strMode: str):
    parts = strMode.split("-")
    assert parts[0] in ["sum", "avg", "linear", "soft"]

    if strMode == "sum":
        assert tenMetric is None
    if strMode == "avg":
        assert tenMetric is None
    if parts[0] == "linear":
        assert tenMetric is not None
    if parts[0] == "soft":
        assert tenMetric is not None

    if len(parts) == 1:
        eps = 0
    else:
        assert parts[1] in ops._EPS, strMode
        eps = ops._EPS[parts[1]]

    if not tenIn.is_cuda:
        # the reference asserts here too (softsplat.py:420-421)
        raise ops.PgdvsHipError("softsplat: input must be on the GPU (no CPU path)")

    return ops.softsplat_fwd(tenIn, tenFlow, tenMetric, ops._MODES[parts[0]], eps)
