#!/usr/bin/env python3
"""Golden vectors for the softsplat BACKWARD (pgdvs/utils/softsplat.py:430-617).
The reference's backward kernels are cupy/NVRTC strings and cannot run here; what can run is
torch autograd through (i) the vectorised statement of the forward kernel used for the forward
fixtures (make_golden._cpu_splat, differentiable in the input and -- through the bilinear
weights -- in the flow) and (ii) the reference's own torch pre/post-processing in softsplat()
(softsplat.py:294-333), which runs unmodified.  Its gradients are the analytic derivatives the
kernels softsplat_ingrad / softsplat_flowgrad implement."""
import pathlib
import sys
import types

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def main():
    MG._install_stubs()
    import pgdvs.utils.softsplat as SS

    SS.softsplat_func = types.SimpleNamespace(apply=MG._cpu_splat)
    rng = np.random.default_rng(99)
    B, C, H, W = 2, 3, 12, 16
    ten_in = rng.normal(size=(B, C, H, W)).astype(np.float32)
    flow = (rng.normal(size=(B, 2, H, W)) * 2.5).astype(np.float32)
    flow[0, :, 0, :4] = [[30.0], [-20.0]]   # far outside: no corner lands in the image
    flow[1, 0, 5, 5], flow[1, 1, 5, 5] = 0.0, 0.0  # exactly on a pixel centre
    metric = rng.normal(size=(B, 1, H, W)).astype(np.float32)
    gout = rng.normal(size=(B, C, H, W)).astype(np.float32)
    out = {}
    for mode in ["sum", "avg", "linear", "soft"]:
        ti = torch.from_numpy(ten_in).requires_grad_(True)
        tf = torch.from_numpy(flow).requires_grad_(True)
        tm = None
        if mode in ("linear", "soft"):
            tm = torch.from_numpy(np.abs(metric) + 0.1 if mode == "linear" else metric).requires_grad_(True)
        y = SS.softsplat(ti, tf, tm, mode)
        y.backward(torch.from_numpy(gout))
        out[f"{mode}_out"] = y.detach().numpy()
        out[f"{mode}_grad_in"] = ti.grad.numpy()
        out[f"{mode}_grad_flow"] = tf.grad.numpy()
        if tm is not None:
            out[f"{mode}_grad_metric"] = tm.grad.numpy()
    np.savez_compressed(OUT / "softsplat_bwd.npz", ten_in=ten_in, ten_flow=flow, ten_metric=metric, grad_out=gout, **out)
    print(f"  softsplat_bwd.npz {(OUT / 'softsplat_bwd.npz').stat().st_size / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
