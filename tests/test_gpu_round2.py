"""GPU tests added in round 2 (MI355X): dispatch hygiene of the fused GNT kernels (weight caches,
autograd guard, loud fallbacks at the shape boundaries), the aggregation's status word with several
views in flight, and BASELINE.json's configurations at their stated sizes against the oracle."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)
from pgdvs_amd import ops, synth  # noqa: E402

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pgdvs_amd import _lib

    _lib.load()


# ---------------------------------------------------------------- GNT dispatch hygiene
def _gnt_inputs(R=7, S=16, V=5, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, device=DEV, generator=g)  # noqa: E731
    mask = (torch.rand(R, S, V, 1, device=DEV, generator=g) < 0.8).float()
    return r(R, S, V, 35), r(R, S, V, 4), mask, r(R, S, 3), r(R, 3)


def test_gnt_packed_weights_follow_load_state_dict():
    """a checkpoint loaded AFTER a warm-up forward must be the one the fused kernels use
    (round-1 caches were invalidated on device changes only)"""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(0)
    a = GNT(netwidth=64, transformer_depth=2).to(DEV).eval()
    torch.manual_seed(1)
    b = GNT(netwidth=64, transformer_depth=2).to(DEV).eval()
    x = _gnt_inputs()
    with torch.no_grad():
        out_a = a(*x)[0].clone()
        out_b = b(*x)[0].clone()
        assert not torch.allclose(out_a, out_b, atol=1e-3)
        a.load_state_dict(b.state_dict())  # in-place copies: same storage, new contents
        out_a2 = a(*x)[0]
    np.testing.assert_allclose(N(out_a2), N(out_b), rtol=0, atol=1e-6)
    with torch.no_grad():  # a plain in-place update (optimiser step) as well
        a.rgb_fc.bias.add_(0.25)
        out_a3 = a(*x)[0]
    np.testing.assert_allclose(N(out_a3)[:, :3], N(out_b)[:, :3] + 0.25, rtol=0, atol=1e-5)


def test_gnt_grad_enabled_takes_the_autograd_path():
    """with gradients enabled and trainable parameters (train_static_renderer=True) the fused kernels
    -- which return tensors without history -- must step aside"""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(0)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    x = _gnt_inputs()
    out = net(*x)[0]
    assert out.requires_grad
    out[:, :3].sum().backward()
    assert net.rgbfeat_fc[0].weight.grad is not None and float(net.rgbfeat_fc[0].weight.grad.abs().sum()) > 0
    with torch.no_grad():
        fused = net(*x)[0]
    assert not fused.requires_grad
    np.testing.assert_allclose(N(fused), N(out), rtol=1e-4, atol=2e-5)
    for p in net.parameters():  # frozen parameters: nothing to differentiate, fused kernels again
        p.requires_grad_(False)
    assert not ops.needs_autograd(net, x[0])


@pytest.mark.parametrize("S,V", [(ops.GNT_RAY_MAX_SAMPLES + 1, 4), (8, 65)])
def test_gnt_shape_boundaries_are_loud_and_correct(S, V, monkeypatch):
    """one past the fused kernels' limits (samples per ray / source views): the torch branch runs, says so,
    raises under PGDVS_GNT_STRICT, and agrees with the fused result on the part both can compute"""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(0)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    x = _gnt_inputs(R=3, S=S, V=V, seed=2)
    ops._fallback_seen.clear()
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = net(*x)[0]
    msgs = [str(i.message) for i in w if "pgdvs_amd" in str(i.message)]
    assert len(msgs) == 1 and ("pgdvs_gnt_ray_layer" in msgs[0] if S > ops.GNT_RAY_MAX_SAMPLES else "pgdvs_gnt_view_layer" in msgs[0]), msgs
    monkeypatch.setenv("PGDVS_GNT_STRICT", "1")
    from pgdvs_amd._lib import PgdvsHipError

    with torch.no_grad(), pytest.raises(PgdvsHipError):
        net(*x)
    monkeypatch.delenv("PGDVS_GNT_STRICT")
    # all-torch statement of the same forward
    ops._GNT_VIEW_ENABLED = False
    try:
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = net(*x)[0]
    finally:
        ops._GNT_VIEW_ENABLED = True
    np.testing.assert_allclose(N(out), N(ref), rtol=1e-4, atol=3e-5)


# ---------------------------------------------------------------- aggregation status word
def test_checked_count_raises_on_error_status():
    from pgdvs_amd._lib import PgdvsHipError

    assert ops.checked_count(torch.tensor([5], device=DEV), "x") == 5
    with pytest.raises(PgdvsHipError):
        ops.checked_count(torch.tensor([-1], device=DEV), "x")


@pytest.mark.parametrize("H,W,n,K,radius", [(96, 128, 600, 3, 0.2), (130, 70, 900, 2, 0.35)])
def test_points_raster_large_radius_binning_path(H, W, n, K, radius):
    """discs wider than a tile side: the tile box of a point exceeds 2 x 2 tiles, so the binning passes leave the
    LDS-table form (ranks in registers) for the per-run global appends"""
    rng = np.random.default_rng(H + W + n)
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    pts = np.concatenate([rng.uniform(-1.4, 1.4, (n, 2)), rng.uniform(0.3, 3.0, (n, 1))], 1).astype(np.float32)
    pts[: n // 6, 2] = pts[n // 6: 2 * (n // 6), 2][: n // 6]  # depth ties -> (z, id) order
    rgb = rng.random((n, 3), dtype=np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cloud = T(np.concatenate([pts, rgb], 1))
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, K, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, radius, K)
    assert (idx >= 0).mean() > 0.5
    assert np.array_equal(N(r["idx"]), idx)
    assert np.array_equal(N(r["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32))
    np.testing.assert_allclose(N(r["rgb"]), orc.composite(idx, d2, radius, rgb), rtol=0, atol=1e-6)


def test_points_raster_more_tiles_than_the_binning_table():
    """2064 x 2064 pixels = 129 x 129 tiles > 16 384 LDS counters: the binning passes count and append through
    global memory; fragments checked on windows against the oracle's naive rasteriser over all points"""
    H = W = 2064
    n, K, radius = 6000, 3, 0.004
    rng = np.random.default_rng(77)
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    z = rng.uniform(0.8, 3.0, (n, 1))
    pts = np.concatenate([rng.uniform(-0.62, 0.62, (n, 2)) * z, z], 1).astype(np.float32)
    # clusters inside the checked windows so that they hold many (and overlapping) discs
    for k, (cy, cx) in enumerate([(0.55, 0.55), (0.0, 0.0), (-0.55, -0.55)]):
        m = slice(k * 1500, (k + 1) * 1500)
        pts[m, 0] = (cx + rng.uniform(-0.03, 0.03, 1500)) * pts[m, 2]
        pts[m, 1] = (cy + rng.uniform(-0.03, 0.03, 1500)) * pts[m, 2]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cloud = T(np.concatenate([pts, rng.random((n, 3), dtype=np.float32)], 1))
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, K, H, W, want_fragments=True)
    ndc = orc.points_to_ndc(pts, fc, H, W)
    gi = N(r["idx"])
    hits = 0
    ys, xs = np.nonzero((gi[..., 0] >= 0))
    assert ys.size > 1000
    # windows around covered pixels spread over the image (first, middle, last in raster order) + a corner
    picks = [(int(ys[i]), int(xs[i])) for i in (0, ys.size // 3, ys.size // 2, 2 * ys.size // 3, ys.size - 1)] + [(0, 0)]
    for (cy, cx) in picks:
        y0, x0 = max(0, min(H - 96, cy - 48)), max(0, min(W - 96, cx - 48))
        idx, zbuf, d2 = orc.rasterize_points_window(ndc, H, W, radius, K, y0, y0 + 96, x0, x0 + 96)
        assert np.array_equal(gi[y0:y0 + 96, x0:x0 + 96], idx), (y0, x0)
        assert np.array_equal(N(r["zbuf"])[y0:y0 + 96, x0:x0 + 96].view(np.uint32), zbuf.view(np.uint32)), (y0, x0)
        assert np.array_equal(N(r["dist2"])[y0:y0 + 96, x0:x0 + 96].view(np.uint32), d2.view(np.uint32)), (y0, x0)
        hits += int((idx >= 0).sum())
    assert hits > 500
    # every covered pixel of the whole image names a point whose disc really covers it (cheap global sanity)
    assert int((gi >= n).sum()) == 0


def test_static_aggregation_packed_xyz_and_raster_from_it():
    """pgdvs_static_aggregate_packed: same cloud, plus the coordinates alone; the rasteriser fed with the packed
    coordinates (stride 3) and the colours of the rows (stride 6) gives the same fragments bit for bit, through
    ops and through the renderer's optional ``st_pcl_xyz``"""
    from types import SimpleNamespace

    from pgdvs_amd.renderers.st_geo_renderer import StaticGeoPointRenderer

    v = synth.make_video(4, 72, 128, seed=5)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    args = (T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]).view(torch.uint8), v["K3s"], v["c2ws"])
    cloud, cnt = ops.static_aggregate(*args)
    cloud2, cnt2, xyz = ops.static_aggregate(*args, return_xyz=True)
    n = int(cnt.item())
    assert int(cnt2.item()) == n and torch.equal(cloud[:n], cloud2[:n]) and torch.equal(xyz[:n], cloud[:n, :3])
    d = synth.make_view(v, 1, seed=3)
    fc = T(d["flat_cam_tgt"][0])
    cam = ops.cam_prep(fc)
    a = ops.points_raster(cloud, cloud[:, 3:], cam, 0.01, 3, 72, 128, n_points_dev=cnt, want_fragments=True)
    b = ops.points_raster(xyz, cloud[:, 3:], cam, 0.01, 3, 72, 128, n_points_dev=cnt, want_fragments=True)
    for k in ("idx", "zbuf", "dist2", "rgb", "mask"):
        assert torch.equal(a[k], b[k]), k
    rc = SimpleNamespace(st_pcl_remove_outlier=False, st_render_pcl_pt_radius=0.01, st_render_pcl_pts_per_pixel=3)
    ren = StaticGeoPointRenderer()
    r0 = ren(tgt_h=72, tgt_w=128, flat_tgt_cam=fc, st_pcl_rgb=cloud, render_cfg=rc, n_points_dev=cnt, planar=True)
    r1 = ren(tgt_h=72, tgt_w=128, flat_tgt_cam=fc, st_pcl_rgb=cloud, render_cfg=rc, n_points_dev=cnt, planar=True,
             st_pcl_xyz=xyz)
    assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])


def test_static_aggregation_ordered_selection_path():
    """frames >= 1 normally run the step chain (one launch per frame: unordered selection bits + stamps; rows built
    at the end).  PGDVS_AGG_ORDERED=1 runs round 2's chain instead -- per frame an ordered selection and a push that
    builds the rows -- an independent second implementation (read once per process, hence the child process)
    through the same bit-exact aggregation tests."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, PGDVS_AGG_ORDERED="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_parity.py"), "-k",
                        "static_aggregation_shapes_vs_oracle or static_aggregation_vs_reference_golden or "
                        "static_aggregation_capacity_clamp"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_static_aggregation_with_views_in_flight():
    """three aggregations in flight on their own streams beside a chip-filling kernel (the benchmark's
    situation): ticketed tile ids make the ordered-offset look-back independent of dispatch order; the
    status word stays clean and every cloud is the oracle's"""
    v = synth.make_video(6, 270, 480, seed=21)  # 16 tiles of 8192 pixels
    rg, de, mk = T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]).view(torch.uint8)
    o = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    filler = torch.randn(4096, 4096, device=DEV)
    results = []
    for it in range(12):
        s = streams[it % 3]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            results.append(ops.static_aggregate(rg, de, mk, v["K3s"], v["c2ws"]))
        filler = filler @ filler * 1e-4  # keeps CUs busy on the default stream
    torch.cuda.synchronize()
    for cloud, cnt in results:
        n = ops.checked_count(cnt, "pgdvs_static_aggregate")
        assert n == o.shape[0]
        assert np.array_equal(N(cloud[:n]).view(np.uint32), o.view(np.uint32))


def test_points_raster_rejects_more_than_2e31_list_entries():
    from pgdvs_amd import _lib
    from pgdvs_amd._lib import PgdvsHipError

    lib = _lib.load()
    assert lib.pgdvs_points_raster_workspace_bytes(1 << 30, 1080, 1920, 0.5) == -4  # PGDVS_ERR_UNSUPPORTED
    with pytest.raises(PgdvsHipError):
        ops._ws(-4, DEV)


def test_points_raster_device_count_is_clamped_to_capacity():
    """a device-side count larger than the rows the workspace was sized for must not be trusted"""
    rng = np.random.default_rng(0)
    H, W, n = 40, 56, 500
    pts = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(1.0, 2.0, (n, 1)), rng.random((n, 3))], 1).astype(np.float32)
    cam = ops.cam_prep(T(synth.flat_cam(H, W, np.array([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]]), np.eye(4)).astype(np.float32)))
    ref = ops.points_raster(T(pts), T(pts)[:, 3:], cam, 0.05, 3, H, W, want_fragments=True)
    big = ops.points_raster(T(pts), T(pts)[:, 3:], cam, 0.05, 3, H, W, want_fragments=True,
                            n_points_dev=torch.tensor([10 * n], dtype=torch.int64, device=DEV))
    neg = ops.points_raster(T(pts), T(pts)[:, 3:], cam, 0.05, 3, H, W, want_fragments=True,
                            n_points_dev=torch.tensor([-1], dtype=torch.int64, device=DEV))
    assert torch.equal(ref["idx"], big["idx"]) and torch.equal(ref["rgb"], big["rgb"])
    assert int((neg["idx"] >= 0).sum()) == 0 and float(neg["mask"].sum()) == 0.0


# ---------------------------------------------------------------- BASELINE.json configurations at their stated sizes
def _renderer(static="geo", **over):
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

    cfg = load_config(static_renderer=static)
    rc = cfg.engine.engine_cfg.render_cfg
    for k, v in over.items():
        rc[k] = v
    return PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(DEV).eval(), rc


def _aggregate_and_check(v, S, H, W):
    """A12 at this size: HIP cloud == oracle cloud, row for row and bit for bit (point ids are the
    rasteriser's tie-break, so order matters)"""
    cloud, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=S * H * W)
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    n = ops.checked_count(cnt, "pgdvs_static_aggregate")
    assert n == o_cloud.shape[0]
    assert np.array_equal(N(cloud[:n]).view(np.uint32), o_cloud.view(np.uint32))
    return cloud, cnt, o_cloud


def _windows(H, W, dyn_mask, size=64):
    """four size x size pixel windows: image corner, image centre, bottom-right edge, and one centred on
    dynamic content (the static render there shows what the dynamic splat composites over)"""
    ys, xs = np.nonzero(dyn_mask)
    cy, cx = (int(ys.mean()), int(xs.mean())) if ys.size else (H // 3, W // 3)
    clampw = lambda y, x: (max(0, min(H - size, y)), max(0, min(W - size, x)))  # noqa: E731
    return [clampw(0, 0), clampw(H // 2 - size // 2, W // 2 - size // 2), clampw(H - size, W - size), clampw(cy - size // 2, cx - size // 2)]


def _check_static_windows(cloud_np, flat_cam_tgt, H, W, radius, K, frag, img_chw, mask_hw, size=64, dyn_mask=None):
    """HIP z-buffer fragments and composite on cropped windows against the oracle's naive rasteriser run
    over ALL points for those pixels (O(window x N): seconds even at 1080p x 3.5 M points), then over the
    whole frame against the oracle's point-major sweep"""
    ndc = orc.points_to_ndc(cloud_np[:, :3], flat_cam_tgt, H, W)
    hits = 0
    for (y0, x0) in _windows(H, W, dyn_mask if dyn_mask is not None else np.zeros((H, W), bool), size):
        y1, x1 = min(H, y0 + size), min(W, x0 + size)
        idx, zbuf, d2 = orc.rasterize_points_window(ndc, H, W, radius, K, y0, y1, x0, x1)
        assert np.array_equal(N(frag["idx"])[y0:y1, x0:x1], idx), (y0, x0)        # integer z-buffer index path: bit-exact
        assert np.array_equal(N(frag["zbuf"])[y0:y1, x0:x1].view(np.uint32), zbuf.view(np.uint32)), (y0, x0)
        assert np.array_equal(N(frag["dist2"])[y0:y1, x0:x1].view(np.uint32), d2.view(np.uint32)), (y0, x0)
        img = orc.composite(idx, d2, radius, cloud_np[:, 3:])
        ones = orc.composite(idx, d2, radius, None)
        np.testing.assert_allclose(N(img_chw)[:, y0:y1, x0:x1], img.transpose(2, 0, 1), rtol=0, atol=1e-6)
        assert np.array_equal(N(mask_hw)[y0:y1, x0:x1], (ones[..., 0] > 0).astype(np.float32))
        hits += int((idx >= 0).sum())
    assert hits > 0
    # ... and EVERY pixel of the frame against the oracle's point-major sweep (the same arithmetic and insertion rule
    # as the naive loop, proven equal to it on the CPU: tests/test_oracle_golden.py::test_pointmajor_raster_equals_naive)
    idx, zbuf, d2 = orc.rasterize_points_pointmajor(ndc, H, W, radius, K)
    assert np.array_equal(N(frag["idx"]), idx)
    assert np.array_equal(N(frag["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(frag["dist2"]).view(np.uint32), d2.view(np.uint32))
    img = orc.composite(idx, d2, radius, cloud_np[:, 3:])
    ones = orc.composite(idx, d2, radius, None)
    np.testing.assert_allclose(N(img_chw), img.transpose(2, 0, 1), rtol=0, atol=1e-6)
    assert np.array_equal(N(mask_hw), (ones[..., 0] > 0).astype(np.float32))


def _full_view_check(v, d, cloud, cnt, o_cloud, H, W, K, radius, remove_outlier=True, st_outlier=False, window=64):
    """whole per-view path on HIP; the static image is pinned on windows against the full-cloud oracle
    raster, then the oracle runs the dynamic branch + composite over the FULL frame on top of it"""
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=remove_outlier, st_render_pcl_pts_per_pixel=K, st_render_pcl_pt_radius=radius)
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"], data["st_pcl_rgb_count"] = cloud[None], cnt
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    n = o_cloud.shape[0]
    cam = ops.cam_prep(data["flat_cam_tgt"][0])
    frag = ops.points_raster(cloud[:n], cloud[:n, 3:], cam, radius, K, H, W, want_fragments=True, rgb_planar=True)
    assert torch.equal(frag["rgb"], ret["geo_static_rgb"][0]) and torch.equal(frag["mask"], ret["geo_static_mask"][0, 0])
    _check_static_windows(o_cloud, d["flat_cam_tgt"][0], H, W, radius, K, frag, ret["geo_static_rgb"][0], ret["geo_static_mask"][0, 0],
                          size=window, dyn_mask=d["dyn_mask_src_temporal"][0, 0, ..., 0] > 0)
    od = dict(d)
    od["rgb_gnt"] = N(ret["geo_static_rgb"]).transpose(0, 2, 3, 1)  # the (window-pinned) static image, [B,H,W,3]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    assert float(ret["render_dyn_mask"].mean()) > 0.03
    for k in ["render_dyn_rgb", "combined_rgb"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)
    assert float(np.mean((N(ret["combined_rgb"]) - o["combined_rgb"]) ** 2)) < 1e-8  # >= 80 dB
    return ret


def test_config_c1_256x256_4_frames_vs_oracle():
    """BASELINE.json configs[0] at its stated size: single 256 x 256 target view, 4 source frames -- every
    stage against the CPU oracle over the full frame (naive O(pixels x points) rasteriser included)"""
    H, W, S = 256, 256, 4
    v = synth.make_video(S, H, W, seed=1234)
    d = synth.make_view(v, 1, frac=0.4, seed=5)
    cloud, cnt, o_cloud = _aggregate_and_check(v, S, H, W)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"], data["st_pcl_rgb_count"] = cloud[None], cnt
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["st_pcl_rgb"] = o_cloud[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(ret["geo_static_mask"]), o["geo_static_mask"])
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    for k in ["geo_static_rgb", "render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)
    frag = ops.points_raster(cloud[:o_cloud.shape[0]], None, ops.cam_prep(data["flat_cam_tgt"][0]), rc.st_render_pcl_pt_radius, 3, H, W,
                             want_fragments=True, want_rgb=False)
    idx, zbuf, d2 = orc.rasterize_points(o_cloud[:, :3], d["flat_cam_tgt"][0], H, W, rc.st_render_pcl_pt_radius, 3)
    assert np.array_equal(N(frag["idx"]), idx) and np.array_equal(N(frag["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(frag["dist2"]).view(np.uint32), d2.view(np.uint32))


def test_config_c3_1080p_24_frames_vs_oracle():
    """BASELINE.json configs[2] -- the benchmark's own workload (1080p, 24 source frames, ~3.5 M static
    points, ~310 k kNN queries) -- end to end: aggregated cloud bit-exact and in order; z-buffer fragments
    bit-exact on four 64 x 64 windows (one over dynamic content) against the oracle's naive loop over ALL points
    for those pixels AND on all 2.07 M pixels against its point-major sweep; dynamic splat (outlier filter on: brute-force kNN in the oracle) + composite over
    the full frame within 1e-4"""
    H, W, S = 1080, 1920, 24
    v = synth.make_video(S, H, W, seed=1234)
    d = synth.make_view(v, 7, frac=0.4, seed=5)
    cloud, cnt, o_cloud = _aggregate_and_check(v, S, H, W)
    assert 1.2 * H * W < o_cloud.shape[0] < 3.0 * H * W
    _full_view_check(v, d, cloud, cnt, o_cloud, H, W, 3, 0.01)


def test_config_c4_nvidia_288x550_rank_slice_vs_oracle():
    """BASELINE.json configs[3]: NVIDIA-Dynamic-Scenes-sized sequence (24 frames at 288 x 550), the views one
    rank of eight renders (DistributedSampler slice of the 12 x 24 = 288 target views), with the settings of
    the reference's pure-geometry benchmark `st_cvd_pcl_clean_dy_cvd_pcl_clean` (scripts/benchmark.sh:90-106:
    st_pcl_remove_outlier=true, knn 50, std_thres 0.2, dyn_pcl_remove_outlier=true, radius 0.01, 3 points
    per pixel) -- full frame against the oracle, statistical filter of the static cloud included"""
    from pgdvs_amd.dist import shard_indices

    H, W, S = 288, 550, 24
    v = synth.make_video(S, H, W, seed=77)
    cloud, cnt, o_cloud = _aggregate_and_check(v, S, H, W)
    mine = shard_indices(12 * 24, rank=3, world=8)
    assert len(mine) == 36 and mine[:3] == [3, 11, 19]
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_pcl_remove_outlier=True, st_pcl_outlier_knn=50,
                          st_pcl_outlier_std_thres=0.2, st_render_pcl_pt_radius=0.01, st_render_pcl_pts_per_pixel=3)
    # The oracle's statistical filter of the static cloud (brute-force kNN over 280 k points: 6 s) does not depend on
    # the view: the first view runs it inside the oracle's renderer, the other 35 get the cloud it kept (the HIP
    # renderer filters in every forward, as the reference does)
    avg = orc.knn_mean_dist(o_cloud[:, :3], 50)
    o_kept = o_cloud[avg < orc.outlier_threshold(avg, 0.2)]
    rc_nofilter = dict(rc, st_pcl_remove_outlier=False)
    for n_view, view in enumerate(mine):  # (time, camera) of the view: frame = view // 12; all 36 views of the rank
        i = min(view // 12, S - 2)
        d = synth.make_view(v, i, frac=0.25 + 0.05 * (view % 12) / 12, seed=view)
        data = synth.to_torch(d, DEV)
        data["st_pcl_rgb"], data["st_pcl_rgb_count"] = cloud[None], cnt
        with torch.no_grad():
            ret = model.forward(data, render_cfg=rc)
        od = dict(d)
        od["st_pcl_rgb"] = (o_cloud if n_view == 0 else o_kept)[None]
        o = orc.render_view(od, dict(rc) if n_view == 0 else rc_nofilter, static_noise=d["static_noise"], alpha=100.0)
        assert np.array_equal(N(ret["geo_static_mask"]), o["geo_static_mask"])
        assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
        for k in ["geo_static_rgb", "render_dyn_rgb", "combined_rgb"]:
            np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=f"view {view}: {k}")


def test_config_c5_1080p_48_frames_vs_oracle():
    """BASELINE.json configs[4]: 1080p, 48 source frames, dynamic-mask compositing (one rank's view): cloud
    bit-exact vs the oracle at S = 48, fragments on windows, dynamic branch + composite over the full frame"""
    H, W, S = 1080, 1920, 48
    v = synth.make_video(S, H, W, seed=4321)
    d = synth.make_view(v, 30, frac=0.6, seed=9)
    cloud, cnt, o_cloud = _aggregate_and_check(v, S, H, W)
    ret = _full_view_check(v, d, cloud, cnt, o_cloud, H, W, 3, 0.01, window=48)
    m = N(ret["render_dyn_mask"])[0, 0] > 0
    # dynamic-mask compositing: inside the mask the view shows the splat, outside the static render
    np.testing.assert_array_equal(N(ret["combined_rgb"])[0][:, ~m], N(ret["geo_static_rgb"])[0][:, ~m])
    np.testing.assert_array_equal(N(ret["combined_rgb"])[0][:, m], N(ret["render_dyn_rgb"])[0][:, m])


# ---------------------------------------------------------------- evaluator-shaped caller (row 8f-1)
def test_eval_step_drives_the_hip_renderer():
    """pgdvs_amd.harness.eval_step (the reference evaluator's step) around the HIP PGDVSRenderer: host-side
    data dict in, device transfer, forward, quantisation, masked PSNRs.  Ground truth = the oracle's image of
    the same view plus a known offset, so the expected PSNRs follow from the oracle alone."""
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl
    from pgdvs_amd.harness import eval_step, masked_psnr, quantize_like_evaluator

    H, W, S = 72, 128, 4
    v = synth.make_video(S, H, W, seed=21)
    d = synth.make_view(v, 1, seed=3)
    cloud = aggregate_static_pcl(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"])
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=20, st_render_pcl_pts_per_pixel=3,
                          st_render_pcl_pt_radius=0.02)
    od = dict(d)
    od["st_pcl_rgb"] = N(cloud)[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    gt = np.clip(o["combined_rgb"].transpose(0, 2, 3, 1) + 0.1, 0, 1).astype(np.float32)   # [B,H,W,3]
    dyn = np.repeat(o["render_dyn_mask"].transpose(0, 2, 3, 1), 3, axis=-1).astype(np.float32)
    data = {k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in d.items()}                # HOST tensors, as a DataLoader yields
    data["st_pcl_rgb"] = cloud[None].cpu()
    data["rgb_tgt"], data["eval_mask"] = torch.from_numpy(gt), torch.from_numpy(dyn)
    data["misc"] = [{"scene_id": "synthetic", "tgt_frame_id": 1, "tgt_cam_id": 0}]
    md, extra = eval_step(model, data, rc, device=DEV, return_images=True)
    assert int(md["eval/count"]) == 1 and md["eval/count"].dtype == torch.int64  # (single process: host tensors)
    pq = quantize_like_evaluator(torch.from_numpy(o["combined_rgb"][0]))
    gq = quantize_like_evaluator(torch.from_numpy(gt[0]).permute(2, 0, 1))
    # quantised predictions: the HIP image is within 1e-4 of the oracle's, so at most a handful of 8-bit steps differ
    # (compared as 8-bit codes: the final division by 255 is not correctly rounded on the GPU)
    b8 = lambda t: (t * 255).round().to(torch.uint8)  # noqa: E731
    assert float((b8(extra["pred"][0].cpu()) != b8(pq)).float().mean()) < 2e-3
    m = torch.from_numpy(dyn[0]).permute(2, 0, 1)
    for key, mask in (("psnr_full_combined", torch.ones_like(m)), ("psnr_dyn_combined", m), ("psnr_static_combined", 1 - m)):
        want = masked_psnr(gq, pq, mask)
        assert abs(float(md[f"eval/{key}"]) - want) < 0.02, (key, float(md[f"eval/{key}"]), want)
    assert 19.0 < float(md["eval/psnr_full_combined"]) < 21.5  # a 0.1 offset is ~20 dB


# ---------------------------------------------------------------- rasteriser: list growth paths
@pytest.mark.parametrize("n,spread,K", [(3000, 0.02, 3), (12000, 0.01, 3), (9000, 0.0, 2)])
def test_points_raster_crowded_tiles_vs_oracle(n, spread, K):
    """tile lists beyond the sorted path's LDS capacity (2560 entries: general path), far beyond it, and
    thousands of points at one and the same depth (one z-bucket: the exact-rank pass hands over to the
    general path) -- all bit-exact vs the oracle"""
    rng = np.random.default_rng(n)
    H, W = 40, 56
    xy = rng.normal(0.0, 0.05, (n, 2))
    z = 2.0 + rng.normal(0.0, spread, n) if spread > 0 else np.full(n, 2.0)
    pts = np.concatenate([xy * z[:, None], z[:, None]], 1).astype(np.float32)
    feat = rng.random((n, 3)).astype(np.float32)
    fc = synth.flat_cam(H, W, np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]]), np.eye(4))
    r = ops.points_raster(T(pts), T(feat), ops.cam_prep(T(fc)), 0.06, K, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, 0.06, K)
    assert (idx >= 0).mean() > 0.03 and int(N(r["idx"]).max()) > n // 2
    assert np.array_equal(N(r["idx"]), idx)
    assert np.array_equal(N(r["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32))
    np.testing.assert_allclose(N(r["rgb"]), orc.composite(idx, d2, 0.06, feat), rtol=0, atol=1e-6)


# ---------------------------------------------------------------- A12: where the projection bounds are tight
@pytest.mark.parametrize("motion", ["static", "tiny_shift", "integer_shift", "roll", "projective_K", "behind", "far"])
def test_static_aggregation_degenerate_camera_motions_vs_oracle(motion):
    """camera motions that put re-projected points exactly on (or within rounding of) pixel boundaries: a static
    camera (every point re-projects to x.000.. / x.999..: the fp32 screening form can decide nothing, the fp64 form
    little, the reference operation order decides), a 1e-7 shift, a shift by whole pixels at constant depth, an
    in-plane roll, and intrinsics with a skew term (K not of the sparse form) -- occupancy decisions identical to
    numpy's fp64 projection, cloud bit-exact and in order"""
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl

    S, H, W = 5, 61, 83
    v = synth.make_video(S, H, W, seed=33)
    K3s, c2ws, depths = v["K3s"].copy(), v["c2ws"].copy(), v["depths"].copy()
    for i in range(S):
        c2ws[i] = c2ws[0]
        K3s[i] = K3s[0]
    if motion == "tiny_shift":
        for i in range(S):
            c2ws[i, 0, 3] += 1e-7 * i
    elif motion == "integer_shift":
        depths[:] = 2.0  # fronto-parallel plane: a translation of d/f per pixel shifts the image by whole pixels
        for i in range(S):
            c2ws[i, 0, 3] += 3 * i * 2.0 / K3s[0, 0, 0]
    elif motion == "roll":
        for i in range(S):
            a = 0.01 * i
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
            c2ws[i, :3, :3] = c2ws[0, :3, :3] @ R
    elif motion == "behind":
        for i in range(1, S, 2):  # odd frames look the other way: the cloud lies behind them (no z > 0 test upstream)
            a = np.deg2rad(170.0)
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            c2ws[i, :3, :3] = c2ws[0, :3, :3] @ R
    elif motion == "far":
        depths *= 4000.0  # coordinates of ~1e4: the error bounds scale with max(|x|,|y|,|z|)
        for i in range(S):
            c2ws[i, 0, 3] += 25.0 * i
    elif motion == "projective_K":
        for i in range(S):
            K3s[i, 0, 1] = 0.3 * i  # skew
            c2ws[i, 0, 3] += 0.01 * i
    st = N(aggregate_static_pcl(T(v["rgbs"]), T(depths), T(v["dyn_masks"]), K3s, c2ws))
    o = orc.aggregate_static_pcl(v["rgbs"], depths, v["dyn_masks"], K3s, c2ws)
    assert st.shape == o.shape, (st.shape, o.shape)
    assert np.array_equal(st.view(np.uint32), o.view(np.uint32))
    if motion == "static":  # dedup is total where the frames agree: only what differs between frames is added
        assert st.shape[0] < 2.2 * H * W


def _bench_line_and_detail(r):
    """(the compact line = the LAST line of stdout, the detail record from stderr)"""
    import json

    out_lines = r.stdout.rstrip("\n").splitlines()
    assert out_lines and out_lines[-1].startswith("{"), r.stdout[-1500:]  # nothing follows the line on stdout
    assert len([ln for ln in out_lines if ln.startswith("{")]) == 1, r.stdout[-1500:]
    assert len(out_lines[-1].encode()) <= 4096, len(out_lines[-1])
    det = [ln for ln in r.stderr.splitlines() if ln.startswith("bench detail: ")]
    assert len(det) == 1
    return json.loads(out_lines[-1]), json.loads(det[0][len("bench detail: "):])


def test_bench_line_contract_small_workload():
    """`bench.py` end to end on the GPU at a small size, GNT leg and scene / configuration sweeps ON: the LAST stdout line is
    one JSON object of at most 4 KB with the driver's fields, the roofline object and the CPU baseline (timed oracle);
    the per-kernel table, the variants and the HIP-vs-oracle check of configs[0] are in the detail record (stderr +
    gpurun_out/bench_detail.json)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--height", "96", "--width", "160", "--frames", "4",
                        "--steps", "12", "--warmup", "3", "--gnt-rays", "128"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    b, det = _bench_line_and_detail(r)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 12 and b["warmup"] == 3 and b["value"] > 0 and b["higher_is_better"] is True
    assert b["vs_baseline"] is None and b["scaling"] == "weak" and b["data"] == "synthetic" and "workload" in b["config"]
    assert abs(b["value"] - 1e3 / b["ms_per_step"]) / b["value"] < 0.02
    rf = b["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes", "avg_ms"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and len(cb["sample"]) <= 200
    assert b["gnt"]["tflops"] > 0 and b["roofline_path"]["frac"] > 0
    # the detail record carries everything the line no longer does
    assert det["value"] == b["value"] and det["cpu_baseline"]["hip_vs_oracle_configs0"]["differing_8bit_values"] == 0
    assert det["kernels"] and det["variants"]["scenes"] and det["variants"]["configs"] and det["gnt"]["parity_vs_torch_statement"]
    on_disk = json.loads(open(os.path.join(root, "gpurun_out", "bench_detail.json")).read())
    assert on_disk["value"] == b["value"]


def test_bench_two_ranks_on_one_gpu():
    """the N-rank code of bench.py's real main (lane count agreed through an all-reduce, per-step asynchronous
    gather of the images to rank 0, barrier + max-over-ranks timing, per-rank rates) with two ranks sharing this
    box's GPU: PGDVS_BENCH_SHARED_GPU_TEST=1 puts every rank on GPU 0 and lets gloo move the device tensors (RCCL
    refuses two ranks on one device); launched the way the driver launches N > 1"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PGDVS_BENCH_SHARED_GPU_TEST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--height", "96",
                        "--width", "160", "--frames", "4", "--steps", "10", "--warmup", "3", "--gnt-rays", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    b, det = _bench_line_and_detail(r)  # rank 0 alone prints
    assert b["n_gpus"] == 2 and b["steps"] == 10 and b["scaling"] == "weak" and b["value"] > 0
    per_rank = b["config"]["per_rank_frames_per_s"]
    assert len(per_rank) == 2 and all(x > 0 for x in per_rank)
    assert b["config"]["gather_bytes_to_rank0"] == 3 * 96 * 160 * 4 * 10
    assert abs(b["value"] - 2 * 1e3 / b["ms_per_step"]) / b["value"] < 0.02  # whole-job rate: both ranks' views / max time
