#!/usr/bin/env python3
"""Golden vectors for the evaluator-shaped caller by RUNNING THE REFERENCE's own
``PGDVSEvaluator.eval_step`` (pgdvs/engines/evaluator_pgdvs.py:26-188) and ``obtain_quantitative_nvidia``
(:190-283) in the build container: the engine object is created without its disk-touching constructor,
its model is a stand-in that returns a fixed ``combined_rgb`` (so the fixture pins the CALLER: to-device,
clamp, NaN handling, quantisation, the resize of the ground truth to the render size, the three masked
PSNRs and the reduced metric dict), LPIPS / SSIM -- out of scope, third-party networks -- are stubbed to 0.
Stores inputs + outputs in harness_eval_step.npz.  Usage: python tests/golden/make_golden_harness.py"""
import pathlib
import sys
import tempfile
import types

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def main():
    MG._install_stubs()
    from unittest.mock import MagicMock

    for m in ["tensorboard", "torch.utils.tensorboard", "jax", "jax.numpy"]:  # not installed here; unused by eval_step
        sys.modules.setdefault(m, MagicMock())
    import pgdvs.engines.evaluator_pgdvs as EV

    EV.calculate_ssim = lambda *a, **k: 0.0
    torch.distributed.init_process_group("gloo", init_method="tcp://127.0.0.1:29671", rank=0, world_size=1)
    rng = np.random.default_rng(5)
    out = {}
    for tag, (B, H, W, rh, rw) in {"same": (2, 20, 28, 20, 28), "strided": (1, 21, 30, 11, 15)}.items():
        pred = rng.normal(0.5, 0.35, (B, 3, rh, rw)).astype(np.float32)
        pred[0, 1, 2, 3] = np.nan
        gt = rng.normal(0.5, 0.3, (B, H, W, 3)).astype(np.float32)
        mask = (rng.random((B, H, W, 1)) < 0.3).astype(np.float32).repeat(3, axis=-1)

        class Fake(torch.nn.Module):
            def forward(self, data_gpu, render_cfg=None, disable_tqdm=True, for_debug=False):
                assert all(isinstance(v, torch.Tensor) for k, v in data_gpu.items() if k in ("rgb_tgt", "eval_mask"))
                return {"combined_rgb": torch.from_numpy(pred)}

        ev = EV.PGDVSEvaluator.__new__(EV.PGDVSEvaluator)
        tmp = pathlib.Path(tempfile.mkdtemp())
        ev.device = torch.device("cpu")
        ev.model = Fake()
        ev.engine_cfg = types.SimpleNamespace(render_cfg=None, quant_type="nvidia")
        ev.cfg = types.SimpleNamespace(rgb_range="0_1")
        ev.verbose = False
        ev.local_rank = 0
        ev.INFO_DIR, ev.VIS_DIR = str(tmp / "info"), str(tmp / "vis")
        ev.lpips_fn = types.SimpleNamespace(forward=lambda *a, **k: torch.zeros(1))
        data = {
            "rgb_src_temporal": torch.zeros(B, 2, H, W, 3), "rgb_tgt": torch.from_numpy(gt), "eval_mask": torch.from_numpy(mask),
            "seq_ids": torch.arange(B * 3).reshape(B, 3),
            "misc": [{"scene_id": "s", "tgt_frame_id": i, "tgt_cam_id": i} for i in range(B)],
        }
        md = ev.eval_step(data=data, epoch=0, global_step=0, save_individual=False)
        out[f"{tag}_pred"], out[f"{tag}_gt"], out[f"{tag}_mask"] = pred, gt, mask
        for k, v in md.items():
            out[f"{tag}_metric__{k.replace('/', '__')}"] = v.numpy()
    np.savez_compressed(OUT / "harness_eval_step.npz", **out)
    print({k: (v.shape, v.reshape(-1)[:1]) for k, v in out.items() if "metric" in k})


if __name__ == "__main__":
    main()
