#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE ITSELF.

Runs only in the build container, where the upstream tree is mounted read-only at
/root/reference.  It imports the reference's Python modules (never copies them)
under ``sys.modules`` stubs for the dependencies that are not installed here
(cupy, hydra, pytorch3d, trimesh, cv2 ...), calls the reference functions on
small seeded synthetic inputs and stores inputs + outputs as ``.npz`` fixtures
next to this script.  The fixtures are data only; tests/test_oracle_golden.py
replays them against oracle/ (CPU) and tests/test_gpu_parity.py against the HIP
path.

What the stubs stand in for (and therefore what is NOT pinned by these vectors):
  * pytorch3d.ops.knn_points -> exact brute-force kNN on squared L2 (that *is*
    pytorch3d's contract; only tie order of equal distances is unpinned).
  * pgdvs.utils.softsplat.softsplat_func (cupy/CUDA only, asserts on CPU,
    softsplat.py:420-421) -> a vectorised torch scatter written from the kernel
    text softsplat.py:352-402.  The torch pre/post-processing in softsplat()
    (softsplat.py:294-333) runs unmodified.
  * pytorch3d rasteriser / compositor: not executable -> no fixture (A9 unpinned).

Usage:  python tests/golden/make_golden.py
"""
import pathlib
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent


def _install_stubs():
    for m in [
        "cupy", "hydra", "hydra.utils", "omegaconf", "trimesh", "cv2", "pytorch3d",
        "pytorch3d.utils", "pytorch3d.ops", "pytorch3d.renderer", "pytorch3d.structures",
        "skimage", "skimage.metrics", "imageio", "imageio_ffmpeg", "boto3", "botocore",
        "torchvision", "torchvision.utils", "lpips", "sklearn.cluster",
    ]:
        sys.modules[m] = MagicMock()

    def knn_points(p1, p2, K, return_nn=True, **kw):
        d = ((p1[0][:, None, :] - p2[0][None, :, :]) ** 2).sum(-1)  # [N, M]
        k_eff = min(K, d.shape[1])
        dists, idx = torch.topk(d, k_eff, dim=1, largest=False, sorted=True)
        if k_eff < K:
            pad = K - k_eff
            dists = torch.cat([dists, torch.zeros(d.shape[0], pad)], 1)
            idx = torch.cat([idx, torch.zeros(d.shape[0], pad, dtype=idx.dtype)], 1)
        nn = p2[0][idx]
        return dists[None], idx[None], nn[None]

    sys.modules["pytorch3d.ops"].knn_points = knn_points
    # several reference modules create a debug directory at import time
    # (e.g. pgdvs_renderer_dyn.py:19-20); the reference tree is read-only for us
    _mkdir = pathlib.Path.mkdir

    def _guarded_mkdir(self, *a, **k):
        if str(self).startswith(str(REF)):
            return None
        return _mkdir(self, *a, **k)

    pathlib.Path.mkdir = _guarded_mkdir
    sys.modules["pytorch3d"].ops = sys.modules["pytorch3d.ops"]
    sys.path.insert(0, str(REF))


def _cpu_splat(ten_in, ten_flow):
    """Vectorised statement of kernel softsplat_out (softsplat.py:352-402)."""
    B, C, H, W = ten_in.shape
    out = ten_in.new_zeros(ten_in.shape)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for n in range(B):
        fx = xs.float() + ten_flow[n, 0]
        fy = ys.float() + ten_flow[n, 1]
        fin = torch.isfinite(fx) & torch.isfinite(fy)
        fx0 = torch.where(fin, fx, torch.zeros_like(fx))
        fy0 = torch.where(fin, fy, torch.zeros_like(fy))
        nwx = torch.floor(fx0).clamp(-4, W + 4).long()
        nwy = torch.floor(fy0).clamp(-4, H + 4).long()
        corners = [
            (nwx, nwy, (nwx + 1).float() - fx0, (nwy + 1).float() - fy0),
            (nwx + 1, nwy, fx0 - nwx.float(), (nwy + 1).float() - fy0),
            (nwx, nwy + 1, (nwx + 1).float() - fx0, fy0 - nwy.float()),
            (nwx + 1, nwy + 1, fx0 - nwx.float(), fy0 - nwy.float()),
        ]
        # out-of-range floor values were clamped above only to keep the index
        # arithmetic finite; genuinely far-away targets fail the bounds test.
        far = (torch.floor(fx0) < -2) | (torch.floor(fx0) > W) | (torch.floor(fy0) < -2) | (torch.floor(fy0) > H)
        for cx, cy, wx, wy in corners:
            ok = fin & ~far & (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
            lin = (cy * W + cx)[ok]
            w = (wx * wy)[ok]
            for c in range(C):
                out[n, c].view(-1).index_put_((lin,), ten_in[n, c][ok] * w, accumulate=True)
    return out


def _flat_cam(H, W, f, c2w, cx=None, cy=None):
    K = np.eye(4, dtype=np.float64)
    K[0, 0] = K[1, 1] = f
    K[0, 2] = W / 2.0 if cx is None else cx
    K[1, 2] = H / 2.0 if cy is None else cy
    return np.concatenate(([H, W], K.flatten(), np.asarray(c2w, np.float64).flatten())).astype(np.float32)


def _pose(yaw_deg, pitch_deg, t):
    y, p = np.deg2rad(yaw_deg), np.deg2rad(pitch_deg)
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    c2w = np.eye(4)
    c2w[:3, :3] = Ry @ Rx
    c2w[:3, 3] = t
    return c2w


def _synth_dyn_inputs(rng, H, W, same_time=False):
    """Small synthetic (depth, flow, pose, RGB) set in the reference's data layout."""
    f = 0.9 * W
    cam1 = _flat_cam(H, W, f, _pose(1.5, -0.5, [0.00, 0.01, 0.0]))
    cam2 = _flat_cam(H, W, f * 1.02, _pose(-2.0, 0.7, [0.05, 0.00, 0.01]), cx=W / 2 + 0.7, cy=H / 2 - 0.4)
    camt = _flat_cam(H, W, f * 0.98, _pose(0.4, 0.3, [0.02, -0.01, -0.02]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = lambda ph: (2.0 + 0.5 * np.sin(xx / W * 3 + ph) + 0.3 * np.cos(yy / H * 2 + ph)).astype(np.float32)
    d1, d2 = depth(0.0), depth(0.4)
    blob = ((xx - W * 0.45) ** 2 + (yy - H * 0.5) ** 2) < (0.3 * min(H, W)) ** 2
    d1[blob] -= 0.8
    d2[blob] -= 0.75
    mask = blob.astype(np.float32)
    mask[1, 1] = 1.0  # isolated pixel -> an outlier
    mask[H - 2, W - 3] = 1.0
    rgb = rng.random((2, H, W, 3), dtype=np.float32)
    flow = (rng.normal(size=(H, W, 2)) * 1.5 + np.array([2.3, -1.2])).astype(np.float32)
    flow[0:3, :, 1] -= 4.0  # some flows leave the image
    occ = (rng.random((H, W, 1)) < 0.1).astype(np.float32)
    t1, t2 = (3.0, 3.0) if same_time else (3.0, 4.0)
    return dict(
        dyn_mask_1=mask[..., None], rgb_1=rgb[0], rgb_2=rgb[1], depth_1=d1[..., None], depth_2=d2[..., None],
        flow_12=flow, flow_12_occ_mask=occ, flat_cam_1=cam1, flat_cam_2=cam2, flat_cam_tgt=camt,
        time_1=np.float32(t1), time_2=np.float32(t2), time_tgt=np.float32(3.4 if not same_time else 3.0),
    )


def main():
    _install_stubs()
    import pgdvs.renderers.pgdvs_renderer_base as RB
    import pgdvs.renderers.pgdvs_renderer_dyn as RD
    import pgdvs.renderers.pgdvs_renderer as RR
    import pgdvs.models.gnt.projector as PJ
    import pgdvs.utils.softsplat as SS
    from pgdvs.models.gnt.renderer import BaseRenderer as GNTRenderer

    SS.softsplat_func = types.SimpleNamespace(apply=_cpu_splat)
    T = torch.from_numpy
    rng = np.random.default_rng(1234)

    # ---------------- A1: get_batched_rays -------------------------------
    base = RB.PGDVSBaseRenderer()
    for name, (H, W, stride) in {"rays_a": (12, 20, 1), "rays_b": (17, 23, 2)}.items():
        fc = _flat_cam(H, W, 0.9 * W, _pose(3.0, -2.0, [0.1, -0.2, 0.05]), cx=W / 2 + 0.3)
        K = T(fc[2:18].reshape(1, 4, 4))
        c2w = T(fc[18:34].reshape(1, 4, 4))
        ro, rd, uvs, _, shape = base.get_batched_rays(
            device="cpu", batch_size=1, H=H, W=W, render_stride=stride, intrinsics=K, c2w=c2w)
        np.savez_compressed(OUT / f"{name}.npz", flat_cam=fc, H=H, W=W, stride=stride,
                            rays_o=ro.numpy(), rays_d=rd.numpy(), uvs=uvs.numpy(), render_hw=np.array(shape))

    # ---------------- A5: Projector.compute_projections -------------------
    proj = PJ.Projector()
    fc = _flat_cam(30, 40, 35.0, _pose(5.0, 3.0, [0.2, 0.1, -0.3]))
    xyz = (rng.normal(size=(200, 1, 3)) * np.array([1.0, 1.0, 2.0]) + np.array([0, 0, 1.5])).astype(np.float32)
    uv, msk = proj.compute_projections(T(xyz), T(fc[None]))
    np.savez_compressed(OUT / "project.npz", flat_cam=fc, xyz=xyz[:, 0], uv=uv[0, :, 0].numpy(), mask=msk[0, :, 0].numpy())

    # ---------------- A2-A5: compute_dyn_pcl ------------------------------
    cfg_ns = types.SimpleNamespace(rgb_range="0_1", tracker=None)
    dyn = RD.PGDVSDynamicRenderer(cfg=cfg_ns, softsplat_metric_abs_alpha=100.0, proj_func=proj.compute_projections)
    H, W = 24, 32
    case_id = 0
    for same_time in (False, True):
        for use_fc in (False, True):
            for rm in (False, True):
                inp = _synth_dyn_inputs(rng, H, W, same_time=same_time)
                rc = types.SimpleNamespace(
                    dyn_render_use_flow_consistency=use_fc, dyn_pcl_remove_outlier=rm, dyn_pcl_outlier_knn=8,
                    dyn_pcl_outlier_std_thres=0.1, dyn_render_type="softsplat")
                fcam1 = T(inp["flat_cam_1"])
                ro, rd, uvs, _, _ = dyn.get_batched_rays(
                    device="cpu", batch_size=1, H=H, W=W, render_stride=1,
                    intrinsics=fcam1[2:18].reshape(1, 4, 4), c2w=fcam1[18:34].reshape(1, 4, 4))
                fcam2 = T(inp["flat_cam_2"])
                flow_1_to_tgt, valid_mask, info = dyn.compute_dyn_pcl(
                    dyn_mask_1=T(inp["dyn_mask_1"]), rgb_1=T(inp["rgb_1"]), uvs_1=uvs, ray_o_1=ro, ray_d_1=rd,
                    depth_1=T(inp["depth_1"]), flow_12=T(inp["flow_12"]), flow_12_occ_mask=T(inp["flow_12_occ_mask"]),
                    rgb_2=T(inp["rgb_2"]), depth_2=T(inp["depth_2"]), K_2=fcam2[2:18].reshape(4, 4),
                    c2w_2=fcam2[18:34].reshape(4, 4), flat_cam_tgt=T(inp["flat_cam_tgt"]),
                    time_1=torch.tensor(inp["time_1"]), time_2=torch.tensor(inp["time_2"]),
                    time_tgt=torch.tensor(inp["time_tgt"]), render_cfg=rc)
                np.savez_compressed(
                    OUT / f"dyn_pcl_{case_id}.npz", **inp, use_flow_consistency=use_fc, remove_outlier=rm,
                    outlier_knn=8, outlier_std_thres=0.1, out_flow_1_to_tgt=flow_1_to_tgt.numpy(),
                    out_valid_dyn_mask_1=valid_mask.numpy(), out_pcl=info["pcl"].numpy(),
                    out_pcl_rgbs=info["pcl_rgbs"].numpy(), out_nn_dist_thres=info["pcl_nn_dist_thres"].numpy())
                case_id += 1

    # ---------------- A6: backwarp + L1 metric ----------------------------
    H, W = 20, 28
    rgb1 = rng.random((1, 3, H, W), dtype=np.float32)
    rgb2 = rng.random((1, 3, H, W), dtype=np.float32)
    flow = (rng.normal(size=(1, 2, H, W)) * 3.0).astype(np.float32)
    warp = base.backwarp_for_softsplat_metric(tenIn=T(rgb2), tenFlow=T(flow))
    l1 = torch.nn.functional.l1_loss(T(rgb1), warp, reduction="none").mean(dim=1, keepdim=True)
    np.savez_compressed(OUT / "backwarp_l1.npz", rgb1=rgb1, rgb2=rgb2, flow=flow, warp=warp.numpy(), l1=l1.numpy())

    # ---------------- A7: softsplat() modes (pre/post-processing) ---------
    H, W = 16, 22
    ten_in = rng.random((2, 3, H, W), dtype=np.float32)
    ten_flow = (rng.normal(size=(2, 2, H, W)) * 2.5).astype(np.float32)
    ten_flow[0, 0, 0, 0] = np.inf
    ten_flow[1, 1, 2, 3] = np.nan
    ten_flow[0, :, 5, 5] = [-40.0, 3.0]
    ten_metric = (rng.normal(size=(2, 1, H, W)) * 2.0).astype(np.float32)
    outs = {}
    for mode in ["sum", "avg", "linear", "soft", "soft-zeroeps", "soft-clipeps"]:
        m = None if mode in ("sum", "avg") else T(ten_metric if not mode.startswith("linear") else np.abs(ten_metric) + 0.1)
        outs["out_" + mode.replace("-", "_")] = SS.softsplat(T(ten_in), T(ten_flow), m, mode).numpy()
    np.savez_compressed(OUT / "softsplat_modes.npz", ten_in=ten_in, ten_flow=ten_flow, ten_metric=ten_metric, **outs)

    # ---------------- A8 + A11: PGDVSRenderer.forward (rgb_gnt shortcut) ---
    def _forward_case(name, rm, use_fc, stride, rng):
        H, W, B = 24, 32, 2
        data = {}
        per = [_synth_dyn_inputs(rng, H, W, same_time=(b == 1 and name == "forward_b")) for b in range(B)]
        per[1]["dyn_mask_1"] = per[1]["dyn_mask_1"] if name == "forward_b" else np.zeros_like(per[1]["dyn_mask_1"])
        data["rgb_src_temporal"] = np.stack([np.stack([p["rgb_1"], p["rgb_2"]]) for p in per])
        data["depth_src_temporal"] = np.stack([np.stack([p["depth_1"], p["depth_2"]]) for p in per])
        data["dyn_mask_src_temporal"] = np.stack([np.stack([p["dyn_mask_1"], p["dyn_mask_1"]]) for p in per])
        data["flow_fwd"] = np.stack([p["flow_12"] for p in per])
        data["flow_fwd_occ_mask"] = np.stack([p["flow_12_occ_mask"] for p in per])
        data["flat_cam_tgt"] = np.stack([p["flat_cam_tgt"] for p in per])
        data["flat_cam_src_temporal"] = np.stack([np.stack([p["flat_cam_1"], p["flat_cam_2"]]) for p in per])
        data["time_tgt"] = np.stack([[p["time_tgt"]] for p in per]).astype(np.float32)
        data["time_src_temporal"] = np.stack([[p["time_1"], p["time_2"]] for p in per]).astype(np.float32)
        data["rgb_gnt"] = rng.random((B, -(-H // stride), -(-W // stride), 3), dtype=np.float32)
        rc = types.SimpleNamespace(
            render_stride=stride, pure_gnt=False, pure_gnt_with_dyn_mask=False, gnt_use_dyn_mask=False,
            gnt_use_masked_spatial_src=False, dyn_render_use_flow_consistency=use_fc, dyn_pcl_remove_outlier=rm,
            dyn_pcl_outlier_knn=8, dyn_pcl_outlier_std_thres=0.1, dyn_render_type="softsplat",
            dyn_render_track_temporal="none")
        model = RR.PGDVSRenderer.__new__(RR.PGDVSRenderer)
        torch.nn.Module.__init__(model)
        model.cfg = cfg_ns
        model.flag_debug = False
        st = GNTRenderer.__new__(GNTRenderer)
        torch.nn.Module.__init__(st)
        st.projector = proj
        model.static_renderer = st
        model.softsplat_metric_abs_alpha = 100.0
        model.dyn_renderer = RD.PGDVSDynamicRenderer(
            cfg=cfg_ns, softsplat_metric_abs_alpha=100.0, proj_func=proj.compute_projections)
        tdata = {k: T(v) for k, v in data.items()}
        tdata["depth_range"] = torch.tensor([[0.5, 5.0]] * B)
        tdata["rgb_src_spatial"] = tdata["rgb_src_temporal"]
        tdata["dyn_mask_src_spatial"] = tdata["dyn_mask_src_temporal"]
        tdata["flat_cam_src_spatial"] = tdata["flat_cam_src_temporal"]
        # record the actual draw of torch.randn_like(rgb_src_1) (pgdvs_renderer_dyn.py:181)
        torch.manual_seed(77)
        drawn = []
        real_randn_like = torch.randn_like

        def _recording_randn_like(t, *a, **k):
            r = real_randn_like(t, *a, **k)
            drawn.append(r.clone())
            return r

        torch.randn_like = _recording_randn_like
        try:
            with torch.no_grad():
                ret = model.forward(tdata, render_cfg=rc, disable_tqdm=True)
        finally:
            torch.randn_like = real_randn_like
        assert len(drawn) == 1 and tuple(drawn[0].shape) == (B, 3, H, W)
        noise = drawn[0].contiguous()
        np.savez_compressed(
            OUT / f"{name}.npz", **{"in_" + k: v for k, v in data.items()}, static_noise=noise.numpy(),
            remove_outlier=rm, use_flow_consistency=use_fc, outlier_knn=8, outlier_std_thres=0.1, render_stride=stride,
            **{"out_" + k: v.numpy() for k, v in ret.items() if torch.is_tensor(v)})

    _forward_case("forward_a", False, False, 1, rng)
    _forward_case("forward_b", True, True, 1, rng)
    # render_stride = 2: bicubic-antialias / nearest resize of the dynamic outputs (:239-248,259-270);
    # own generator so that the fixtures above keep their values
    _forward_case("forward_c", True, False, 2, np.random.default_rng(555))

    # ---------------- A12: static aggregation ------------------------------
    import tempfile
    import PIL.Image
    import pgdvs.datasets.nvidia_eval_pure_geo as PG
    H, W, S = PG.TGT_HEIGHT, 36, 3   # directory name must be images_{W}x288 (nvidia_eval_pure_geo.py:185)
    ds = PG.NvidiaDynPureGeoEvaluationDataset.__new__(PG.NvidiaDynPureGeoEvaluationDataset)
    imgs = rng.integers(0, 256, size=(S, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    depths = np.stack([(2.0 + 0.4 * np.sin(xx / W * 2 + i) + 0.2 * np.cos(yy / H * 5 + 0.3 * i)).astype(np.float32) for i in range(S)])
    masks = np.stack([(((xx - W * (0.3 + 0.1 * i)) ** 2 + (yy - H * 0.5) ** 2) < (0.2 * W) ** 2) for i in range(S)])
    hwf = np.stack([np.array([H, W, 0.9 * W + 0.5 * i], np.float64) for i in range(S)])
    c2ws = np.stack([_pose(1.3 * i, -0.6 * i, [0.03 * i, 0.011 * i, -0.007 * i]) for i in range(S)])
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        mono = td / "scene" / "dense" / f"images_{W}x{H}"
        mono.mkdir(parents=True)
        for i in range(S):
            PIL.Image.fromarray(imgs[i]).save(mono / f"{i:05d}.png")
        ds.raw_data_dir = td
        ds._read_cam = lambda scene_id: (hwf.copy(), c2ws.copy())
        ds._read_depth = lambda scene_id, i: depths[i]
        ds._read_mask = lambda scene_id, i, h, w: masks[i]
        st_pcl_rgb = ds._aggregate_static_pcl("scene")
    # single-step pieces for finer-grained pinning
    K0 = ds._hwf_to_K(hwf[1], normalized=False)
    pcl0 = ds._compute_pcl(H, W, K0, c2ws[1], depths[1])
    pm = ds._compute_pcl_proj_mask(h=H, w=W, pcl=pcl0, K=ds._hwf_to_K(hwf[2], normalized=False),
                                   w2c=np.linalg.inv(c2ws[2]), dyn_mask=masks[2])
    np.savez_compressed(OUT / "static_agg.npz", imgs=imgs, depths=depths, dyn_masks=masks, hwf=hwf, c2ws=c2ws,
                        st_pcl_rgb=st_pcl_rgb, pcl_frame1=pcl0, proj_mask_1_into_2=pm)
    print("golden fixtures written to", OUT)
    for f in sorted(OUT.glob("*.npz")):
        print(f"  {f.name:28s} {f.stat().st_size/1024:8.1f} KiB")


if __name__ == "__main__":
    main()


def make_harness_golden():
    """(pred, gt, mask) -> psnr through the reference's calculate_psnr + the evaluator's
    quantisation expression (evaluator_pgdvs.py:70-77)."""
    _install_stubs()
    from pgdvs.utils.training import calculate_psnr

    rng = np.random.default_rng(77)
    H, W = 20, 30
    pred = rng.normal(0.5, 0.4, (3, H, W)).astype(np.float32)
    pred[0, 0, 0] = np.nan
    gt = rng.random((3, H, W), dtype=np.float32)
    mask = (rng.random((3, H, W)) < 0.7).astype(np.float32)
    p = torch.nan_to_num(torch.from_numpy(pred).clamp(0.0, 1.0), nan=0.0)
    pq = ((p * 255).byte().float() / 255.0).numpy()
    gq = ((torch.from_numpy(gt).clamp(0, 1) * 255).byte().float() / 255.0).numpy()
    np.savez_compressed(OUT / "harness_psnr.npz", pred=pred, gt=gt, mask=mask, pred_q=pq, gt_q=gq,
                        psnr=calculate_psnr(pq.transpose(1, 2, 0), gq.transpose(1, 2, 0), mask.transpose(1, 2, 0)),
                        psnr_same=calculate_psnr(gq.transpose(1, 2, 0), gq.transpose(1, 2, 0), mask.transpose(1, 2, 0)))
