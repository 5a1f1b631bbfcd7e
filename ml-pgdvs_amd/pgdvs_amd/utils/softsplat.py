"""Drop-in for ``pgdvs.utils.softsplat`` (pgdvs/utils/softsplat.py:280-617).

Same signature, modes and assertions.  The cupy/NVRTC kernels are replaced by gfx950 kernels:
``softsplat_out`` by ``pgdvs_softsplat_fwd`` (with the exp / premultiply / normalise steps of
softsplat() fused in when no gradient is needed -- the reference's engines run under
``torch.no_grad()``), ``softsplat_ingrad`` / ``softsplat_flowgrad`` by ``pgdvs_softsplat_bwd``
behind ``softsplat_func`` (a ``torch.autograd.Function`` like upstream's).  When a gradient is
required the pre/post-processing stays in torch around the raw splat, exactly as upstream
(:294-333), so autograd composes the same way.
"""
import torch

from .. import ops


class softsplat_func(torch.autograd.Function):
    """raw summation splat with its analytic backward (softsplat.py:336-617)"""

    @staticmethod
    def forward(ctx, tenIn, tenFlow):
        if not tenIn.is_cuda or not tenFlow.is_cuda:
            raise ops.PgdvsHipError("softsplat: input must be on the GPU (no CPU path)")  # asserts upstream (:420-421)
        ctx.save_for_backward(tenIn, tenFlow)
        return ops.softsplat_fwd(tenIn, tenFlow, None, ops._MODES["sum"], 0)

    @staticmethod
    def backward(ctx, tenOutgrad):
        tenIn, tenFlow = ctx.saved_tensors
        gi, gf = ops.softsplat_bwd(tenIn, tenFlow, tenOutgrad.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gi, gf


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

    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (tenIn, tenFlow, tenMetric))
    if not needs_grad:
        return ops.softsplat_fwd(tenIn, tenFlow, tenMetric, ops._MODES[parts[0]], eps)

    # differentiable path: torch pre/post-processing around the raw splat (:294-333)
    if parts[0] == "avg":
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif parts[0] == "linear":
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif parts[0] == "soft":
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)
    tenOut = softsplat_func.apply(tenIn, tenFlow)
    if parts[0] in ["avg", "linear", "soft"]:
        tenNormalize = tenOut[:, -1:, :, :]
        if len(parts) == 1:
            tenNormalize = tenNormalize + 0.0000001
        elif parts[1] == "addeps":
            tenNormalize = tenNormalize + 0.0000001
        elif parts[1] == "zeroeps":
            tenNormalize = torch.where(tenNormalize == 0.0, torch.ones_like(tenNormalize), tenNormalize)
        elif parts[1] == "clipeps":
            tenNormalize = tenNormalize.clip(0.0000001, None)
        tenOut = tenOut[:, :-1, :, :] / tenNormalize
    return tenOut
