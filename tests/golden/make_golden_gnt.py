#!/usr/bin/env python3
"""Golden vectors for the GNT rows (A13-A16 of SURVEY.md 8a), produced by running the
reference's own modules (pgdvs/models/gnt/*) on the CPU of the build container with a
seeded random-init model.  Same rules as make_golden.py: the reference is imported, never
copied; fixtures are data (inputs, weights of the small net, outputs).

  gnt_small.npz   Projector.compute, sample_along_camera_ray, GNT.forward (2 transformer
                  layers, width 64, 3 source views, 12 samples/ray) incl. the view entropy /
                  std side outputs, render_rays reductions, with and without dynamic masks
  gnt_resunet.npz ResUNet feature maps of a seeded random-init network (the 35 MB of
                  weights are not stored: the mirror network must reproduce torch's init
                  sequence under the same seed)
  gnt_render.npz  BaseRenderer.forward end to end (feature net + chunk loop) on a tiny view
"""
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
    from pgdvs.models.gnt.model import GNTModel
    from pgdvs.models.gnt.projector import Projector
    from pgdvs.models.gnt.ray_sampler import sample_along_camera_ray
    from pgdvs.models.gnt.renderer import BaseRenderer
    import pgdvs.renderers.pgdvs_renderer_base as RB

    T = torch.from_numpy
    rng = np.random.default_rng(2024)
    H, W, V, Ss = 32, 48, 3, 12
    torch.manual_seed(123)
    model = GNTModel(netwidth=64, transformer_depth=2, coarse_feat_dim=32, fine_feat_dim=32, single_net=True,
                     posenc_max_freq_log2=9, pos_enc_n_freqs=10, view_enc_n_freqs=10).eval()
    # make LayerNorm / bias parameters non-trivial so that mistakes show
    with torch.no_grad():
        for n, p in model.net_coarse.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.1)

    f = 0.9 * W
    cams_src = np.stack([MG._flat_cam(H, W, f * (1 + 0.03 * i), MG._pose(3.0 * i - 3, 1.0 * i, [0.15 * i - 0.15, 0.02 * i, 0.01 * i]),
                                      cx=W / 2 + 0.3 * i) for i in range(V)])
    cam_tgt = MG._flat_cam(H, W, f, MG._pose(0.5, -0.4, [0.03, -0.02, 0.0]))
    src_rgbs = rng.random((1, V, H, W, 3), dtype=np.float32)
    inv_masks = (rng.random((1, V, H, W, 1)) < 0.25).astype(np.float32)
    depth_range = np.array([[0.8, 4.0]], np.float32)

    # ---- ResUNet -----------------------------------------------------------------
    with torch.no_grad():
        feat_c, feat_f = model.feature_net(T(src_rgbs[0]).permute(0, 3, 1, 2))
    np.savez_compressed(OUT / "gnt_resunet.npz", seed=123, src_rgbs=src_rgbs, feat=feat_c.numpy(),
                        n_params=sum(p.numel() for p in model.feature_net.parameters()))

    # ---- rays of the target view (reference's own generator) -----------------------
    base = RB.PGDVSBaseRenderer()
    ro, rd, uvs, brefs, _ = base.get_batched_rays(
        device="cpu", batch_size=1, H=H, W=W, render_stride=1,
        intrinsics=T(cam_tgt[2:18].reshape(1, 4, 4)), c2w=T(cam_tgt[18:34].reshape(1, 4, 4)))
    sel = torch.from_numpy(rng.permutation(H * W)[:40])
    ro, rd = ro[sel], rd[sel]
    dr = T(depth_range)[torch.zeros(len(sel), dtype=torch.long)]
    pts, z_vals = sample_along_camera_ray(ro, rd, dr, Ss, inv_uniform=True, det=True)

    proj = Projector()
    outs = {}
    for tag, masks in (("nomask", None), ("dynmask", T(inv_masks))):
        with torch.no_grad():
            pr = proj.compute(xyz=pts, query_camera=T(cam_tgt[None]), train_imgs=T(src_rgbs), train_cameras=T(cams_src[None]),
                              featmaps=feat_c, train_invalid_masks=masks)
            rgb, extra = model.net_coarse(pr["rgb_feat"], pr["ray_diff"], pr["mask"], pts, rd,
                                          ret_view_entropy=True, ret_view_std=True)
        outs.update({
            f"{tag}_rgb_feat": pr["rgb_feat"].numpy(), f"{tag}_ray_diff": pr["ray_diff"].numpy(),
            f"{tag}_mask_inbound": pr["mask_inbound"].numpy(), f"{tag}_mask": pr["mask"].numpy(),
            f"{tag}_out": rgb.numpy(), f"{tag}_view_entropy": extra["view_entropy"].numpy(),
            f"{tag}_view_std": extra["view_std"].numpy(), f"{tag}_view_std_normalized": extra["view_std_normalized"].numpy()})
        if masks is not None:
            outs[f"{tag}_mask_invalid"] = pr["mask_invalid"].numpy()
    weights = {"w_" + k: v.numpy() for k, v in model.net_coarse.state_dict().items()}
    np.savez_compressed(OUT / "gnt_small.npz", H=H, W=W, V=V, Ss=Ss, cams_src=cams_src, cam_tgt=cam_tgt, src_rgbs=src_rgbs,
                        inv_masks=inv_masks, depth_range=depth_range, featmaps=feat_c.numpy(), ray_o=ro.numpy(),
                        ray_d=rd.numpy(), pts=pts.numpy(), z_vals=z_vals.numpy(), **outs, **weights)

    # ---- BaseRenderer.forward end to end --------------------------------------------
    br = BaseRenderer.__new__(BaseRenderer)
    torch.nn.Module.__init__(br)
    br.projector = proj
    br.model = model
    ro, rd, uvs, brefs, shape = base.get_batched_rays(
        device="cpu", batch_size=1, H=H, W=W, render_stride=2,
        intrinsics=T(cam_tgt[2:18].reshape(1, 4, 4)), c2w=T(cam_tgt[18:34].reshape(1, 4, 4)))
    ray_batch = {
        "ray_o": ro, "ray_d": rd, "camera": T(cam_tgt[None]), "rgb": None, "batch_refs": brefs, "view_uv": uvs,
        "raw_h": H, "raw_w": W, "render_h": shape[0], "render_w": shape[1], "depth_range": T(depth_range),
        "depth_range_per_ray": False, "src_rgbs": T(src_rgbs), "src_invalid_masks": T(inv_masks),
        "src_cameras": T(cams_src[None])}
    with torch.no_grad():
        ret = br.forward(ray_batch=ray_batch, chunk_size=37, inv_uniform=True, n_coarse_samples_per_ray=Ss,
                         n_fine_samples_per_ray=0, use_dyn_mask=True, flag_deterministic=True, render_stride=2,
                         ret_view_entropy=True, ret_view_std=True, disable_tqdm=True)
    # importance re-sampling: 6 extra samples per ray drawn from the coarse weights, second pass
    with torch.no_grad():
        ret_f = br.forward(ray_batch=ray_batch, chunk_size=53, inv_uniform=True, n_coarse_samples_per_ray=Ss,
                           n_fine_samples_per_ray=6, use_dyn_mask=True, flag_deterministic=True, render_stride=2,
                           ret_view_entropy=True, ret_view_std=True, disable_tqdm=True)
    np.savez_compressed(OUT / "gnt_render.npz", render_stride=2, chunk_size=37,
                        **{"out_" + k: v.numpy() for k, v in ret["outputs_coarse"].items()},
                        fine_chunk_size=53, n_fine=6,
                        **{"fine_" + k: v.numpy() for k, v in ret_f["outputs_fine"].items()},
                        **{"finec_" + k: v.numpy() for k, v in ret_f["outputs_coarse"].items()})
    for fn in ("gnt_small.npz", "gnt_resunet.npz", "gnt_render.npz"):
        print(f"  {fn:20s} {(OUT / fn).stat().st_size / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
