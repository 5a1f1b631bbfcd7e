"""GPU parity tests (MI355X): the HIP path, called through the C ABI, against
  (a) the CPU oracle on the same seeded inputs,
  (b) the committed golden vectors produced by the reference itself,
  (c) size-independent properties at BASELINE.json's full size (1080p).
Integer / index outputs are compared bit-exactly; float outputs within the stated
tolerances (north_star: 1e-4 on [0,1] images)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)
from pgdvs_amd import ops, synth  # noqa: E402
from pgdvs_amd.instantiate import AttrDict, load_config  # noqa: E402
from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer  # noqa: E402
from pgdvs_amd.utils.softsplat import softsplat  # noqa: E402

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().cpu().numpy()


def _load(golden_dir, name):
    return dict(np.load(golden_dir / name))


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pgdvs_amd import _lib

    _lib.load()  # fails loudly if the HIP extension is missing


# ---------------------------------------------------------------- cameras / rays
def test_cam_prep_bit_exact(golden_dir):
    g = _load(golden_dir, "dyn_pcl_0.npz")
    for k in ("flat_cam_1", "flat_cam_2", "flat_cam_tgt"):
        blk = N(ops.cam_prep(T(g[k])))
        assert np.array_equal(blk.view(np.uint32), orc.cam_prep(g[k]).view(np.uint32)), k


@pytest.mark.parametrize("name", ["rays_a.npz", "rays_b.npz"])
def test_rays(golden_dir, name):
    g = _load(golden_dir, name)
    H, W, s = int(g["H"]), int(g["W"]), int(g["stride"])
    ro, rd, uv, shape = ops.get_rays(ops.cam_prep(T(g["flat_cam"])), H, W, s)
    o_ro, o_rd, o_uv, o_shape = orc.get_batched_rays(g["flat_cam"], H, W, s)
    assert tuple(shape) == tuple(o_shape) == tuple(g["render_hw"])
    assert np.array_equal(N(uv), o_uv) and np.array_equal(N(ro), o_ro)
    assert np.array_equal(N(rd).view(np.uint32), o_rd.view(np.uint32))  # same op order: bit-exact
    np.testing.assert_allclose(N(rd), g["rays_d"], rtol=2e-6, atol=2e-6)  # vs the reference


# ---------------------------------------------------------------- compaction
@pytest.mark.parametrize("n", [0, 1, 63, 4096, 4097, 100003, 2073600])
def test_compact(n):
    rng = np.random.default_rng(n)
    flags = (rng.random(n) < 0.3).astype(np.uint8) * rng.integers(1, 255, n).astype(np.uint8)
    idx, cnt = ops.compact_u8(T(flags))
    ref = np.flatnonzero(flags)
    assert int(cnt.item()) == ref.size
    assert np.array_equal(N(idx)[: ref.size], ref.astype(np.int32))


# ---------------------------------------------------------------- dyn branch pieces
def _dyn_case(g):
    cams = ops.cam_prep(T(np.stack([g["flat_cam_1"], g["flat_cam_2"], g["flat_cam_tgt"]])))
    times = T(np.array([g["time_1"], g["time_2"], g["time_tgt"]], np.float32))
    return cams, times


@pytest.mark.parametrize("case", range(8))
def test_compute_dyn_pcl_vs_oracle_and_golden(golden_dir, case):
    g = _load(golden_dir, f"dyn_pcl_{case}.npz")
    cams, times = _dyn_case(g)
    rc = AttrDict(dyn_render_use_flow_consistency=bool(g["use_flow_consistency"]),
                  dyn_pcl_remove_outlier=bool(g["remove_outlier"]), dyn_pcl_outlier_knn=int(g["outlier_knn"]),
                  dyn_pcl_outlier_std_thres=float(g["outlier_std_thres"]))
    from pgdvs_amd.renderers.pgdvs_renderer_dyn import PGDVSDynamicRenderer

    dyn = PGDVSDynamicRenderer(cfg=AttrDict(rgb_range="0_1"), proj_func=None)
    flow, vmask, info = dyn.compute_dyn_pcl(
        dyn_mask_1=T(g["dyn_mask_1"][..., 0]), rgb_1=T(g["rgb_1"]), depth_1=T(g["depth_1"][..., 0]),
        flow_12=T(g["flow_12"]), flow_12_occ_mask=T(g["flow_12_occ_mask"][..., 0]), rgb_2=T(g["rgb_2"]),
        depth_2=T(g["depth_2"][..., 0]), cam_1=cams[0], cam_2=cams[1], cam_tgt=cams[2], times=times,
        render_cfg=rc, need_points=True)
    o = orc.compute_dyn_pcl(
        dyn_mask_1=g["dyn_mask_1"], rgb_1=g["rgb_1"], depth_1=g["depth_1"], flow_12=g["flow_12"],
        flow_12_occ_mask=g["flow_12_occ_mask"], rgb_2=g["rgb_2"], depth_2=g["depth_2"], flat_cam_1=g["flat_cam_1"],
        flat_cam_2=g["flat_cam_2"], flat_cam_tgt=g["flat_cam_tgt"], time_1=float(g["time_1"]), time_2=float(g["time_2"]),
        time_tgt=float(g["time_tgt"]), dyn_render_use_flow_consistency=rc.dyn_render_use_flow_consistency,
        dyn_pcl_remove_outlier=rc.dyn_pcl_remove_outlier, dyn_pcl_outlier_knn=rc.dyn_pcl_outlier_knn,
        dyn_pcl_outlier_std_thres=rc.dyn_pcl_outlier_std_thres)
    H, W = g["dyn_mask_1"].shape[:2]
    # integer paths: bit-exact vs oracle AND vs the reference
    assert np.array_equal(N(info["valid"]).astype(bool), o["valid"])
    assert np.array_equal(N(vmask), o["valid_dyn_mask_1"][..., 0])
    assert np.array_equal(N(vmask), g["out_valid_dyn_mask_1"][..., 0])
    # float paths: same op order as the oracle -> bit-exact; vs reference within tolerance
    vb = o["valid"]
    assert np.array_equal(N(info["pcl_dense"])[vb].view(np.uint32), o["pcl_dense"][vb].view(np.uint32))
    f_gpu = N(flow).transpose(1, 2, 0)
    assert np.array_equal(f_gpu.view(np.uint32), o["flow_1_to_tgt"].view(np.uint32))
    np.testing.assert_allclose(f_gpu, g["out_flow_1_to_tgt"], rtol=1e-4, atol=2e-4)
    n = int(info["n_pts"].item())
    assert n == g["out_pcl"].shape[0]
    np.testing.assert_allclose(N(info["pcl"])[:n], g["out_pcl"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(N(info["pcl_rgbs"])[:n], g["out_pcl_rgbs"], rtol=1e-5, atol=1e-5)
    if rc.dyn_pcl_remove_outlier:
        nv = int(info["n_valid"].item())
        assert np.array_equal(N(info["avg_nn_dist"])[:nv].view(np.uint32), o["avg_nn_dist"].view(np.uint32))
        np.testing.assert_allclose(N(info["pcl_nn_dist_thres"])[0], g["out_nn_dist_thres"], rtol=1e-5)
        np.testing.assert_allclose(N(info["pcl_nn_dist_thres"])[0], o["pcl_nn_dist_thres"], rtol=1e-6)


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("n,K", [(1, 4), (5, 8), (51, 50), (300, 50), (2000, 50), (5000, 16)])
def test_knn_mean_dist_and_threshold(n, K, algo):
    rng = np.random.default_rng(n * 131 + K)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    pts[: n // 10] = pts[n // 10: 2 * (n // 10)][: n // 10]  # duplicates -> distance ties at 0
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    avg = ops.knn_mean_dist(T(pts), cnt, K, algo=algo)
    ref = orc.knn_mean_dist(pts, K)
    assert np.array_equal(N(avg)[:n].view(np.uint32), ref.view(np.uint32))
    thres, flag = ops.outlier_flags(avg, cnt, 0.1, True)
    o_thres = orc.outlier_threshold(ref, 0.1)
    if n > 1:
        np.testing.assert_allclose(N(thres)[0], o_thres, rtol=1e-6)
        margin = np.abs(ref - o_thres) > 1e-5 * abs(o_thres)
        assert np.array_equal(N(flag)[:n].astype(bool)[margin], (ref < o_thres)[margin])
    else:
        assert np.isnan(N(thres)[0])


@pytest.mark.parametrize("kind", ["surface", "clusters", "line", "identical"])
def test_knn_grid_exact_on_hard_distributions(kind):
    """the grid search must return exactly the brute-force values: surface samples with
    flying-pixel outliers (ring expansion + fallback scan), tight clusters far apart, a
    degenerate line, and all-identical points."""
    rng = np.random.default_rng(sum(map(ord, kind)))
    n, K = 12000, 50
    if kind == "surface":
        u = rng.uniform(-1, 1, (n, 2))
        pts = np.stack([u[:, 0], u[:, 1], 2 + 0.3 * np.sin(3 * u[:, 0])], 1)
        pts[:150, 2] += rng.uniform(0.2, 3.0, 150)          # outliers off the surface
        pts[150:160] = [50.0, -30.0, 9.0]                    # far cluster smaller than K
    elif kind == "clusters":
        c = rng.uniform(-100, 100, (40, 3))
        pts = c[rng.integers(0, 40, n)] + rng.normal(0, 1e-3, (n, 3))
    elif kind == "line":
        t = rng.uniform(0, 1, n)
        pts = np.stack([t, 2 * t, -t], 1)
    else:
        pts = np.ones((n, 3))
    pts = pts.astype(np.float32)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    a_grid = N(ops.knn_mean_dist(T(pts), cnt, K, algo=2))[:n]
    ref = orc.knn_mean_dist(pts, K)
    assert np.array_equal(a_grid.view(np.uint32), ref.view(np.uint32))
    # capacity larger than the device-side count: trailing rows must be ignored
    pad = np.concatenate([pts, rng.normal(size=(500, 3)).astype(np.float32)])
    a_pad = N(ops.knn_mean_dist(T(pad), cnt, K, algo=2))[:n]
    assert np.array_equal(a_pad.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("K", [8, 20, 50])
def test_knn_search_variants_agree_at_benchmark_size(K, monkeypatch):
    """two depth-map-like sheets of 150k points each (the 1080p filter input is 311k): the
    thread-per-query first pass + ring search must give bit-identical means to the pure
    wavefront-per-query ring search (itself pinned against the oracle above)"""
    rng = np.random.default_rng(K)
    n = 300_000
    u = rng.uniform(-1, 1, (n, 2))
    z = 2 + 0.3 * np.sin(3 * u[:, 0]) + np.where(np.arange(n) % 2 == 0, 0.0, 0.004)
    pts = np.stack([u[:, 0] * z, u[:, 1] * z * 0.6, z], 1)
    pts[:400] += rng.normal(0, 0.5, (400, 3))  # flying pixels
    pts = pts.astype(np.float32)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    with ops.option("knn_no_tpq", 1):
        a0 = N(ops.knn_mean_dist(T(pts), cnt, K, algo=2))
    a1 = N(ops.knn_mean_dist(T(pts), cnt, K, algo=2))
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
    assert np.isfinite(a1).all() and float(a1.max()) > 10 * float(np.median(a1))


def test_backwarp_l1(golden_dir):
    g = _load(golden_dir, "backwarp_l1.npz")
    l1 = N(ops.backwarp_l1(T(g["rgb1"]), T(g["rgb2"]), T(g["flow"])))
    np.testing.assert_allclose(l1, g["l1"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(l1[0, 0], orc.backwarp_l1(g["rgb1"][0], g["rgb2"][0], g["flow"][0]), rtol=0, atol=1e-7)


# ---------------------------------------------------------------- softsplat op
@pytest.mark.parametrize("mode", ["sum", "avg", "linear", "soft", "soft-zeroeps", "soft-clipeps"])
def test_softsplat_modes(golden_dir, mode):
    g = _load(golden_dir, "softsplat_modes.npz")
    metric = None if mode in ("sum", "avg") else (g["ten_metric"] if mode != "linear" else np.abs(g["ten_metric"]) + 0.1)
    out = N(softsplat(T(g["ten_in"]), T(g["ten_flow"]), None if metric is None else T(metric), mode))
    np.testing.assert_allclose(out, g["out_" + mode.replace("-", "_")], rtol=2e-5, atol=2e-6)  # vs reference
    np.testing.assert_allclose(out, orc.softsplat(g["ten_in"], g["ten_flow"], metric, mode), rtol=2e-5, atol=2e-6)


def test_softsplat_backward_vs_oracle_and_autograd_golden(golden_dir):
    g = _load(golden_dir, "softsplat_bwd.npz")
    # raw kernels: bit-exact against the oracle restatement
    gi, gf = ops.softsplat_bwd(T(g["ten_in"]), T(g["ten_flow"]), T(g["grad_out"]))
    o_gi, o_gf = orc.softsplat_bwd_raw(g["ten_in"], g["ten_flow"], g["grad_out"])
    assert np.array_equal(N(gi).view(np.uint32), o_gi.view(np.uint32))
    assert np.array_equal(N(gf).view(np.uint32), o_gf.view(np.uint32))
    # every mode through autograd vs the reference wrapper's autograd
    for mode in ["sum", "avg", "linear", "soft"]:
        ti = T(g["ten_in"]).requires_grad_(True)
        tf = T(g["ten_flow"]).requires_grad_(True)
        tm = None
        if mode in ("linear", "soft"):
            tm = T(np.abs(g["ten_metric"]) + 0.1 if mode == "linear" else g["ten_metric"]).requires_grad_(True)
        y = softsplat(ti, tf, tm, mode)
        y.backward(T(g["grad_out"]))
        # forward of the differentiable path == fused inference path
        with torch.no_grad():
            y_fused = softsplat(ti.detach(), tf.detach(), None if tm is None else tm.detach(), mode)
        np.testing.assert_allclose(N(y), N(y_fused), rtol=0, atol=1e-5, err_msg=mode)
        np.testing.assert_allclose(N(y), g[f"{mode}_out"], rtol=0, atol=1e-4, err_msg=mode)
        scale = max(1.0, float(np.abs(g[f"{mode}_grad_flow"]).max()))
        np.testing.assert_allclose(N(ti.grad), g[f"{mode}_grad_in"], rtol=0, atol=1e-4, err_msg=mode)
        np.testing.assert_allclose(N(tf.grad), g[f"{mode}_grad_flow"], rtol=0, atol=1e-5 * scale, err_msg=mode)
        if tm is not None:
            np.testing.assert_allclose(N(tm.grad), g[f"{mode}_grad_metric"], rtol=0, atol=1e-4, err_msg=mode)


def test_softsplat_argument_checks():
    x, f = torch.zeros(1, 3, 4, 4, device=DEV), torch.zeros(1, 2, 4, 4, device=DEV)
    with pytest.raises(AssertionError):
        softsplat(x, f, None, "soft")
    with pytest.raises(AssertionError):
        softsplat(x, f, torch.zeros(1, 1, 4, 4, device=DEV), "sum")
    with pytest.raises(AssertionError):
        softsplat(x, f, None, "max")


def test_softsplat_corner_indices_bit_exact():
    """integer path of the splat: which destination texels receive weight."""
    rng = np.random.default_rng(5)
    H, W = 37, 53
    flow = (rng.normal(size=(1, 2, H, W)) * 6).astype(np.float32)
    flow[0, :, 3, 4] = [np.inf, 0]
    flow[0, :, 0, 0] = [-0.0, 0.0]
    flow[0, :, 5, 5] = [1e9, -1e9]
    ones = np.ones((1, 1, H, W), np.float32)
    out = N(softsplat(T(ones), T(flow), None, "sum"))[0, 0]
    idx = orc.softsplat_corners(flow[0])  # [H,W,4]
    ref = orc.softsplat_raw(ones, flow)[0, 0]
    assert np.array_equal(out > 0, ref > 0)
    touched = np.zeros(H * W, bool)
    touched[idx[idx >= 0]] = True
    assert not np.any((out.reshape(-1) > 0) & ~touched)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.sum(), ref.sum(), rtol=1e-5)


# ---------------------------------------------------------------- full forward vs reference
def _renderer(static="gnt", **over):
    cfg = load_config(static_renderer=static)
    rc = cfg.engine.engine_cfg.render_cfg
    for k, v in over.items():
        rc[k] = v
    return PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(DEV).eval(), rc


@pytest.mark.parametrize("name", ["forward_a.npz", "forward_b.npz", "forward_c.npz"])
def test_forward_vs_reference_golden(golden_dir, name):
    """forward_c renders at render_stride 2: the dynamic outputs go through the bicubic-antialias /
    nearest resize of pgdvs_renderer_dyn.py:239-248 before the composite"""
    g = _load(golden_dir, name)
    data = {k[3:]: T(v) for k, v in g.items() if k.startswith("in_")}
    data["static_noise"] = T(g["static_noise"])
    model, rc = _renderer("gnt", render_stride=int(g["render_stride"]),
                          dyn_render_use_flow_consistency=bool(g["use_flow_consistency"]),
                          dyn_pcl_remove_outlier=bool(g["remove_outlier"]), dyn_pcl_outlier_knn=int(g["outlier_knn"]),
                          dyn_pcl_outlier_std_thres=float(g["outlier_std_thres"]))
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc, disable_tqdm=True)
    assert np.array_equal(N(ret["render_dyn_mask"]), g["out_render_dyn_mask"])  # thresholded: exact
    for k in ["render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn", "static_coarse_rgb",
              "render_dyn_temporal_closest_rgb", "render_dyn_temporal_track_rgb"]:
        np.testing.assert_allclose(N(ret[k]), g["out_" + k], rtol=0, atol=1e-4, err_msg=k)
    assert set(k[4:] for k in g if k.startswith("out_")) <= set(ret.keys())


# ---------------------------------------------------------------- rasteriser (A9)
@pytest.mark.parametrize("H,W,n,K,radius", [(24, 32, 300, 1, 0.05), (40, 30, 800, 3, 0.04), (33, 65, 500, 8, 0.08),
                                            (64, 64, 4000, 3, 0.01), (16, 16, 0, 2, 0.05)])
def test_points_raster_vs_oracle(H, W, n, K, radius):
    rng = np.random.default_rng(H * W + n)
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    pts = np.concatenate([rng.uniform(-1.2, 1.2, (n, 2)), rng.uniform(-0.3, 3.0, (n, 1))], 1).astype(np.float32)
    if n:
        pts[: n // 8, 2] = pts[n // 8: 2 * (n // 8), 2][: n // 8]   # equal depths -> (z, idx) tie-break
        pts[:4, :2] = pts[4:8, :2]
    rgb = rng.random((n, 3), dtype=np.float32)
    cloud = T(np.concatenate([pts, rgb], 1)) if n else torch.zeros((0, 6), device=DEV)
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, K, H, W, want_fragments=True)
    if n == 0:
        assert float(r["mask"].abs().sum()) == 0 and int((r["idx"] != -1).sum()) == 0
        return
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, radius, K)
    assert np.array_equal(N(r["idx"]), idx)  # integer z-buffer index path: bit-exact
    assert np.array_equal(N(r["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32))
    img = orc.composite(idx, d2, radius, rgb)
    ones = orc.composite(idx, d2, radius, None)
    np.testing.assert_allclose(N(r["rgb"]), img, rtol=0, atol=1e-6)
    assert np.array_equal(N(r["mask"]), (ones[..., 0] > 0).astype(np.float32))


@pytest.mark.parametrize("K", [1, 3])
def test_points_raster_dense_layers_vs_oracle(K):
    """tens of disc hits per pixel from stacked depth layers (with exact depth ties): every pixel's
    list fills early, so the wave-level depth cull of the tile kernel is active for most of
    each tile's list -- results must still be the oracle's bit for bit"""
    rng = np.random.default_rng(17 + K)
    H, W, n, radius = 48, 56, 30000, 0.07
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    layer = rng.integers(0, 6, n)
    z = 1.0 + 0.15 * layer + np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 0.02, n))  # half of them tie exactly
    xy = rng.uniform(-1.3, 1.3, (n, 2)) * z[:, None] * 0.6
    pts = np.concatenate([xy, z[:, None]], 1).astype(np.float32)
    order = rng.permutation(n)  # far layers arrive before near ones as often as not
    pts = pts[order]
    rgb = rng.random((n, 3), dtype=np.float32)
    cloud = T(np.concatenate([pts, rgb], 1))
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, K, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, radius, K)
    assert float((idx[..., K - 1] >= 0).mean()) > 0.95  # lists are full almost everywhere
    assert np.array_equal(N(r["idx"]), idx)
    assert np.array_equal(N(r["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32))


def test_points_raster_device_count_and_planar():
    rng = np.random.default_rng(3)
    H, W, n = 30, 44, 600
    fc = synth.flat_cam(H, W, *synth.frame_camera(0, 4, H, W))
    cloud = T(np.concatenate([rng.uniform(-1, 1, (n, 2)), rng.uniform(0.5, 3, (n, 1)), rng.random((n, 3))], 1).astype(np.float32))
    cam = ops.cam_prep(T(fc))
    a = ops.points_raster(cloud[:400].contiguous(), cloud[:400, 3:], cam, 0.03, 3, H, W)
    cnt = torch.tensor([400], dtype=torch.int64, device=DEV)
    b = ops.points_raster(cloud, cloud[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, rgb_planar=True)
    assert torch.equal(a["rgb"].permute(2, 0, 1), b["rgb"]) and torch.equal(a["mask"], b["mask"])


# ---------------------------------------------------------------- static aggregation (A12)
def test_static_aggregation_vs_reference_golden(golden_dir):
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl, hwf_to_K

    g = _load(golden_dir, "static_agg.npz")
    S, H, W = g["depths"].shape
    K3s = np.stack([hwf_to_K(*g["hwf"][i]) for i in range(S)])
    rgbs = g["imgs"].astype(np.float32) / 255.0
    st = N(aggregate_static_pcl(T(rgbs), T(g["depths"]), T(g["dyn_masks"]), K3s, g["c2ws"]))
    assert st.shape == g["st_pcl_rgb"].shape  # same occupancy decisions as the reference
    np.testing.assert_allclose(st[:, :3], g["st_pcl_rgb"][:, :3], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[:, 3:], g["st_pcl_rgb"][:, 3:], rtol=0, atol=1e-6)
    o = orc.aggregate_static_pcl(rgbs, g["depths"], g["dyn_masks"], K3s, g["c2ws"])
    assert np.array_equal(st.view(np.uint32), o.view(np.uint32))  # vs oracle: bit-exact


def test_static_aggregation_vs_oracle_synth():
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl

    v = synth.make_video(5, 54, 96, seed=11)
    st = N(aggregate_static_pcl(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"]))
    o = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    assert st.shape == o.shape
    assert np.array_equal(st.view(np.uint32), o.view(np.uint32))
    # dedup must bite: far fewer points than S*P
    assert st.shape[0] < 0.6 * 5 * 54 * 96


def test_pure_geo_dataset_static_cloud_vs_reference(golden_dir, tmp_path):
    """on-disk sequence -> NvidiaDynPureGeoEvaluationDataset mirror (HIP aggregation at
    construction) vs what the reference's dataset class produced on the same tree"""
    import sys as _sys

    _sys.path.insert(0, str(golden_dir))
    import nvidia_tree as NT
    from pgdvs_amd.datasets.nvidia_eval import NvidiaDynPureGeoEvaluationDataset

    root = NT.build_tree(tmp_path)
    g = _load(golden_dir, "nvidia_items.npz")
    ds = NvidiaDynPureGeoEvaluationDataset(
        data_root=root, raw_data_dir="raw", depth_data_dir="depths", mask_data_dir="masks", flow_data_dir="flows", max_hw=-1,
        mode="eval", scene_ids=[NT.SCENE], flow_consist_thres=1.0, device=DEV)
    item = ds[5 * NT.N_CAMS + 5]
    assert sorted(item.keys()) == list(g["pg_keys"])
    st = item["st_pcl_rgb"].numpy()
    assert st.shape == tuple(g["pg_st_pcl_rgb__shape"])  # same occupancy decisions
    np.testing.assert_allclose(st[:64], g["pg_st_pcl_rgb_head"], rtol=1e-5, atol=1e-6)
    a = st.astype(np.float64).reshape(-1)
    dig = np.array([a @ np.random.default_rng(12345).random(a.size), a.sum(), a.min(), a.max()])
    np.testing.assert_allclose(dig, g["pg_st_pcl_rgb__digest"], rtol=1e-5)
    np.testing.assert_allclose(item["flat_cam_tgt"].numpy(), g["pg_flat_cam_tgt"], rtol=1e-6)


@pytest.mark.parametrize("S,H,W", [(4, 53, 37), (3, 100, 123), (2, 2200, 2000), (1, 60, 84), (5, 60, 84), (6, 47, 61), (9, 60, 84)])
def test_static_aggregation_shapes_vs_oracle(S, H, W):
    """frame sizes that are not multiples of the 16-pixel vector width (unaligned mask rows of
    later frames), a partial last tile, > 1024 tiles (ticketed tile ids), and odd / even frame
    counts (frames are marked in pairs; a single frame has no marking pass at all)"""
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl

    v = synth.make_video(S, H, W, seed=S * 7 + W)
    st = N(aggregate_static_pcl(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"]))
    o = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    assert st.shape == o.shape
    assert np.array_equal(st.view(np.uint32), o.view(np.uint32))


def test_static_aggregation_capacity_clamp():
    v = synth.make_video(3, 54, 96, seed=11)
    full, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]).view(torch.uint8), v["K3s"], v["c2ws"])
    n = int(cnt.item())
    cap = n - 1000
    part, cnt2 = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]).view(torch.uint8), v["K3s"], v["c2ws"],
                                      capacity=cap)
    # overflow: both chains (this test also runs with the option agg_ordered) report count == capacity -- the signal that rows
    # may have been dropped and the cloud must not be used (harness.eval_step and bench.py raise on it); which rows of the
    # later frames survive differs between the chains, frame 0's prefix does not
    assert int(cnt2.item()) == cap
    k = min(cap, 54 * 96 // 2)  # rows appended before the clamp bit are unaffected
    assert torch.equal(part[:k], full[:k])


# ---------------------------------------------------------------- whole view vs oracle (geo static)
@pytest.mark.parametrize("H,W,S,rm,K", [(54, 96, 4, False, 1), (72, 128, 4, True, 3)])
def test_render_view_geo_vs_oracle(H, W, S, rm, K):
    from pgdvs_amd.datasets.static_aggregation import aggregate_static_pcl

    v = synth.make_video(S, H, W, seed=21)
    d = synth.make_view(v, 1, seed=3)
    cloud = aggregate_static_pcl(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"])
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=rm, dyn_pcl_outlier_knn=20, st_render_pcl_pts_per_pixel=K,
                          st_render_pcl_pt_radius=0.02)
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"] = cloud[None]
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["st_pcl_rgb"] = N(cloud)[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(ret["geo_static_mask"]), o["geo_static_mask"])
    np.testing.assert_allclose(N(ret["geo_static_rgb"]), o["geo_static_rgb"], rtol=0, atol=1e-6)
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    assert float(ret["render_dyn_mask"].mean()) > 0.03  # the dynamic discs are actually rendered
    for k in ["render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)
    mse = float(np.mean((N(ret["combined_rgb"]) - o["combined_rgb"]) ** 2))
    assert mse < 1e-8  # PSNR-equivalent bound (>= 80 dB)


def test_config_c2_540p_12_frames_vs_oracle():
    """BASELINE.json configs[1]: 540p target, 12 source frames -- the whole per-view path (static
    aggregation, z-buffer raster K=3, flow-warped splat with the outlier filter, composite) against
    the CPU oracle at the benchmark's own settings (tens of seconds of oracle time)."""
    H, W, S = 540, 960, 12
    v = synth.make_video(S, H, W, seed=1234)
    d = synth.make_view(v, S // 2 - 1, seed=0)
    cloud, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=S * H * W)
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    n = int(cnt.item())
    assert n == o_cloud.shape[0]
    assert np.array_equal(N(cloud[:n]).view(np.uint32), o_cloud.view(np.uint32))  # ordered, bit-exact
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
    for k in ["geo_static_rgb", "render_dyn_rgb", "combined_rgb"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)
    assert float(np.mean((N(ret["combined_rgb"]) - o["combined_rgb"]) ** 2)) < 1e-8


def test_dyn_pcl_render_type_vs_oracle():
    v = synth.make_video(3, 54, 96, seed=5)
    d = synth.make_view(v, 0, seed=1)
    model, rc = _renderer("gnt", dyn_render_type="pcl", dyn_render_pcl_pts_per_pixel=3, dyn_render_pcl_pt_radius=0.03)
    data = synth.to_torch(d, DEV)
    data["rgb_gnt"] = T(v["rgbs"][:1])
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["rgb_gnt"] = v["rgbs"][:1]
    o = orc.render_view(od, dict(rc), static_noise=None)
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    np.testing.assert_allclose(N(ret["combined_rgb"]), o["combined_rgb"], rtol=0, atol=1e-5)


# ---------------------------------------------------------------- A10 mesh variant
@pytest.mark.parametrize("H,W,noise,seed", [(54, 96, 0.0, 0), (96, 54, 0.3, 1), (64, 64, 2.0, 2)])
def test_mesh_render_vs_oracle(H, W, noise, seed):
    """smooth sheet, noisy sheet (stretched / overlapping triangles, z fights) and a wild cloud
    with vertices behind the camera: winning face ids, mask and colours must match exactly"""
    rng = np.random.default_rng(seed)
    v = synth.make_video(2, H, W, seed=seed)
    K3, c2w = v["K3s"][0], v["c2ws"][0]
    cam_src = synth.flat_cam(H, W, K3, c2w)
    pcl = orc.compute_pcl(H, W, K3, c2w, v["depths"][0] + rng.normal(0, 1, (H, W)).astype(np.float32) * noise).reshape(H, W, 3)
    keep = (rng.random((H, W)) < 0.8) | v["dyn_masks"][0]
    keep[0, :3] = [False, True, True]  # the first kept pixel (vertex index 0) is (0,1)
    rgb = v["rgbs"][0]
    cam_tgt = synth.make_view(v, 0, seed=seed)["flat_cam_tgt"][0]
    o_img, o_mask, o_face = orc.mesh_render(keep, pcl, rgb, cam_tgt)
    r = ops.mesh_render(ops.cam_prep(T(cam_tgt)), T(keep.astype(np.uint8)), T(pcl), T(rgb), H, W, want_faces=True)
    assert o_mask.sum() > 0.3 * H * W * (1 if noise < 1 else 0.1)
    assert np.array_equal(N(r["face"]).astype(np.int64), o_face)
    assert np.array_equal(N(r["mask"]), o_mask)
    np.testing.assert_allclose(N(r["rgb"]).transpose(1, 2, 0), o_img, rtol=0, atol=1e-6)
    # no face may use the first kept pixel
    first = np.flatnonzero(keep.reshape(-1))[0]
    used = o_face[o_face >= 0] % (H * W)
    assert not np.any(used == first)


def test_mesh_render_type_end_to_end_vs_oracle():
    v = synth.make_video(3, 54, 96, seed=5)
    d = synth.make_view(v, 0, seed=1)
    model, rc = _renderer("gnt", dyn_render_type="mesh", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=10)
    data = synth.to_torch(d, DEV)
    data["rgb_gnt"] = T(v["rgbs"][:1])
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["rgb_gnt"] = v["rgbs"][:1]
    o = orc.render_view(od, dict(rc), static_noise=None)
    assert o["render_dyn_mask"].mean() > 0.03
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    np.testing.assert_allclose(N(ret["render_dyn_rgb"]), o["render_dyn_rgb"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(N(ret["combined_rgb"]), o["combined_rgb"], rtol=0, atol=1e-5)


def test_mesh_render_540p_vs_oracle():
    """a full 960x540 frame: ~1 M candidate faces, winners / mask / colours exact"""
    H, W = 540, 960
    v = synth.make_video(2, H, W, seed=9)
    d = synth.make_view(v, 0, seed=2)
    K3, c2w = v["K3s"][0], v["c2ws"][0]
    pcl = orc.compute_pcl(H, W, K3, c2w, v["depths"][0]).reshape(H, W, 3)
    keep = v["dyn_masks"][0] | (np.random.default_rng(1).random((H, W)) < 0.3)
    o_img, o_mask, o_face = orc.mesh_render(keep, pcl, v["rgbs"][0], d["flat_cam_tgt"][0])
    r = ops.mesh_render(ops.cam_prep(T(d["flat_cam_tgt"][0])), T(keep.astype(np.uint8)), T(pcl), T(v["rgbs"][0]), H, W,
                        want_faces=True)
    assert o_mask.mean() > 0.1
    assert np.array_equal(N(r["face"]).astype(np.int64), o_face)
    assert np.array_equal(N(r["mask"]), o_mask)
    np.testing.assert_allclose(N(r["rgb"]).transpose(1, 2, 0), o_img, rtol=0, atol=1e-6)


# ---------------------------------------------------------------- A17 tracker-window aggregation
@pytest.mark.parametrize("nq,nb,KK", [(1, 1, 3), (300, 40, 51), (5000, 3000, 51), (2000, 12000, 17), (700, 9000, 64)])
def test_knn_cross_mean_dist_vs_oracle(nq, nb, KK):
    """queries inside, on the border of and far outside the base cloud's bounding box, duplicates
    of base points (distance 0), fewer base points than KK (zero-padded columns)"""
    rng = np.random.default_rng(nq * 7 + nb)
    u = rng.uniform(-1, 1, (nb, 2))
    base = np.stack([u[:, 0], u[:, 1], 2 + 0.3 * np.sin(3 * u[:, 0])], 1).astype(np.float32)
    q = rng.uniform(-1.2, 1.2, (nq, 3)).astype(np.float32)
    q[:, 2] += 2
    q[: nq // 5] = base[rng.integers(0, nb, nq // 5)]
    q[nq // 5: nq // 4] += np.float32(40.0)  # far outside the grid
    qc = torch.tensor([nq], dtype=torch.int32, device=DEV)
    bc = torch.tensor([nb], dtype=torch.int32, device=DEV)
    ref = orc.knn_cross_mean_dist(q, base, KK)
    got = N(ops.knn_cross_mean_dist(T(q), qc, T(base), bc, KK))[:nq]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # capacities larger than the device counts: trailing rows are ignored
    qpad = np.concatenate([q, rng.normal(size=(37, 3)).astype(np.float32)])
    bpad = np.concatenate([base, rng.normal(size=(91, 3)).astype(np.float32)])
    got = N(ops.knn_cross_mean_dist(T(qpad), qc, T(bpad), bc, KK))[:nq]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _track_renderer(**over):
    from pgdvs_amd.renderers.pgdvs_renderer_dyn_track import PGDVSDynamicTrackRenderer

    cfg = load_config(static_renderer="gnt")
    rc = cfg.engine.engine_cfg.render_cfg
    for k, v in over.items():
        rc[k] = v
    return PGDVSDynamicTrackRenderer(cfg=cfg, use_tracker=True).to(DEV), rc


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_track_compute_pcl_for_tgt_vs_reference_and_oracle(golden_dir, case):
    g = _load(golden_dir, "track_pcl.npz")
    raw = {k[5:]: v for k, v in g.items() if k.startswith("data_")}
    rend, rc = _track_renderer(dyn_pcl_outlier_knn=int(g[f"c{case}_knn"]), dyn_pcl_track_track2base_thres_mult=50,
                               dyn_pcl_outlier_std_thres=0.1)
    data = {k: T(v) for k, v in raw.items()}
    dft = rend.prepare_data(0, data, 8, DEV)
    assert dft["idx_temporal_closest"] == list(g["dfk_idx_closest"]) and dft["idx_real_track"] == list(g["dfk_idx_real"])
    assert np.array_equal(N(dft["time_for_track"]), g["dfk_times"]) and np.array_equal(N(dft["time_tgt"]), g["dfk_time_tgt"])
    tracks, vis = g[f"c{case}_tracks"], g[f"c{case}_vis"]
    # per-track stage: bit-exact against the oracle
    odft = orc.track_prepare_data(raw, 0)
    o_valid, o_pcl, o_rgb = orc.track_points(odft, tracks, vis)
    valid, pcl_all, rgb_all = ops.track_points(T(tracks), T(vis), dft["frame_kind"], dft["time_for_track_raw"], dft["time_tgt_raw"],
                                               dft["rgbs_for_track"], dft["depths_for_track"][..., 0], dft["cams_for_track"])
    assert np.array_equal(N(valid).astype(bool), o_valid)
    assert np.array_equal(N(pcl_all).view(np.uint32), o_pcl.view(np.uint32))
    assert np.array_equal(N(rgb_all).view(np.uint32), o_rgb.view(np.uint32))
    # whole row against the reference's output
    wb = bool(g[f"c{case}_with_base"])
    th = g[f"c{case}_base_thres"]
    base = {"pcl": T(g[f"c{case}_base_pts"]) if wb else None, "pcl_rgbs": T(g[f"c{case}_base_rgb"]) if wb else None,
            "pcl_nn_dist_thres": None if np.isnan(th) else T(np.array([th], np.float32))}
    pcl, rgb = rend.compute_pcl_for_tgt(data_for_track=dft, query_pts=T(g[f"c{case}_query"]), tracks=T(tracks),
                                        track_visibles=T(vis), render_cfg=rc, base_pcl_info=base, device=DEV)
    assert tuple(pcl.shape) == g[f"c{case}_out_pcl"].shape  # same filter decisions
    np.testing.assert_allclose(N(pcl), g[f"c{case}_out_pcl"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(N(rgb), g[f"c{case}_out_rgb"], rtol=0, atol=1e-6)
    # base cloud given as a capacity-sized buffer with a device-side count
    if wb:
        nb = g[f"c{case}_base_pts"].shape[0]
        junk = np.full((50, 3), 1.5, np.float32)
        base2 = {"pcl": T(np.concatenate([g[f"c{case}_base_pts"], junk])), "pcl_rgbs": T(np.concatenate([g[f"c{case}_base_rgb"], junk])),
                 "pcl_nn_dist_thres": base["pcl_nn_dist_thres"], "n_pts": torch.tensor([nb], dtype=torch.int32, device=DEV)}
        pcl2, rgb2 = rend.compute_pcl_for_tgt(data_for_track=dft, query_pts=None, tracks=T(tracks), track_visibles=T(vis),
                                              render_cfg=rc, base_pcl_info=base2, device=DEV)
        assert torch.equal(pcl2, pcl) and torch.equal(rgb2, rgb)


@pytest.mark.parametrize("dyn_type", ["softsplat", "pcl"])
def test_render_with_track_end_to_end_vs_oracle(dyn_type):
    v = synth.make_video(7, 54, 96, seed=5)
    d = synth.make_view(v, 3, seed=1)
    synth.add_track_window(d, v, 3, n_side=2)
    over = dict(dyn_render_type=dyn_type, dyn_render_track_temporal="no_tgt", dyn_pcl_outlier_knn=8,
                dyn_render_pcl_pts_per_pixel=3, dyn_render_pcl_pt_radius=0.03)
    model, rc = _renderer("gnt", **over)
    from pgdvs_amd.renderers.pgdvs_renderer_dyn_track import PGDVSDynamicTrackRenderer

    assert isinstance(model.dyn_renderer, PGDVSDynamicTrackRenderer)
    data = synth.to_torch(d, DEV)
    data["rgb_gnt"] = T(v["rgbs"][:1])
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["rgb_gnt"] = v["rgbs"][:1]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"])
    info = o["_info"]
    assert info["temporal_track_mask"].sum() > 100  # the track cloud is actually rendered
    assert np.array_equal(N(ret["render_dyn_temporal_track_mask"]), info["temporal_track_mask"])
    assert np.array_equal(N(ret["render_dyn_temporal_closest_mask"]), info["temporal_closest_mask"])
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    assert o["render_dyn_mask"].sum() > info["temporal_closest_mask"].sum()  # pixels filled from the tracks
    np.testing.assert_allclose(N(ret["render_dyn_temporal_track_rgb"]), info["temporal_track_rgb"], rtol=0, atol=1e-5)
    for k in ["render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)


def test_track_points_large_vs_oracle():
    """~60 k tracks over an 8-frame window at 270x480: per-track stage bit-exact, whole row same
    point set as the oracle"""
    v = synth.make_video(9, 270, 480, seed=4)
    d = synth.make_view(v, 4, seed=1)
    synth.add_track_window(d, v, 4, n_side=3, step=1)
    rend, rc = _track_renderer(dyn_pcl_outlier_knn=20)
    data = synth.to_torch(d, DEV)
    dft = rend.prepare_data(0, data, 8, DEV)
    tracks, vis = d["track_tracks"][0], d["track_visibles"][0]
    assert tracks.shape[0] > 40000
    odft = orc.track_prepare_data(d, 0)
    o_valid, o_pcl, o_rgb = orc.track_points(odft, tracks, vis)
    valid, pcl_all, rgb_all = ops.track_points(T(tracks), T(vis), dft["frame_kind"], dft["time_for_track_raw"], dft["time_tgt_raw"],
                                               dft["rgbs_for_track"], dft["depths_for_track"][..., 0], dft["cams_for_track"])
    assert np.array_equal(N(valid).astype(bool), o_valid) and o_valid.sum() > 5000
    assert np.array_equal(N(pcl_all).view(np.uint32), o_pcl.view(np.uint32))
    assert np.array_equal(N(rgb_all).view(np.uint32), o_rgb.view(np.uint32))
    base = orc.compute_dyn_pcl(
        dyn_mask_1=d["dyn_mask_src_temporal"][0, 0], rgb_1=d["rgb_src_temporal"][0, 0], depth_1=d["depth_src_temporal"][0, 0],
        flow_12=d["flow_fwd"][0], flow_12_occ_mask=d["flow_fwd_occ_mask"][0], rgb_2=d["rgb_src_temporal"][0, 1],
        depth_2=d["depth_src_temporal"][0, 1], flat_cam_1=d["flat_cam_src_temporal"][0, 0],
        flat_cam_2=d["flat_cam_src_temporal"][0, 1], flat_cam_tgt=d["flat_cam_tgt"][0], time_1=float(d["time_src_temporal"][0, 0]),
        time_2=float(d["time_src_temporal"][0, 1]), time_tgt=float(d["time_tgt"][0, 0]), dyn_pcl_outlier_knn=20)
    o_pcl2, o_rgb2, _ = orc.track_compute_pcl_for_tgt(odft, tracks, vis, dict(rc), base["pcl"], base["pcl_rgbs"], base["pcl_nn_dist_thres"])
    binfo = {"pcl": T(base["pcl"]), "pcl_rgbs": T(base["pcl_rgbs"]), "pcl_nn_dist_thres": T(np.array([base["pcl_nn_dist_thres"]], np.float32))}
    pcl2, rgb2 = rend.compute_pcl_for_tgt(data_for_track=dft, query_pts=None, tracks=T(tracks), track_visibles=T(vis),
                                          render_cfg=rc, base_pcl_info=binfo, device=DEV)
    assert tuple(pcl2.shape) == o_pcl2.shape and o_pcl2.shape[0] > base["pcl"].shape[0]
    assert np.array_equal(N(pcl2).view(np.uint32), o_pcl2.view(np.uint32))
    assert np.array_equal(N(rgb2).view(np.uint32), o_rgb2.view(np.uint32))


def test_track_renderer_requires_tracks_or_tracker():
    v = synth.make_video(7, 27, 48, seed=5)
    d = synth.make_view(v, 3, seed=1)
    synth.add_track_window(d, v, 3, n_side=2)
    d.pop("track_tracks")
    model, rc = _renderer("gnt", dyn_render_track_temporal="no_tgt", dyn_pcl_outlier_knn=8)
    data = synth.to_torch(d, DEV)
    data["rgb_gnt"] = T(v["rgbs"][:1])
    with pytest.raises(KeyError, match="track_tracks"):
        model.forward(data, render_cfg=rc)


# ---------------------------------------------------------------- full-size properties (1080p)
def test_fullsize_properties_1080p():
    """BASELINE size: properties that need no oracle run.
    - zero flow + constant metric: soft splat is the identity (in / (1 + 1e-7/e)).
    - 'sum' splat conserves mass for interior flows (bilinear weights sum to 1).
    - rasterising a cloud sampled on the pixel grid of the target camera itself returns every
      point at its own pixel (idx == pixel id where K=1 and radius < half a pixel)."""
    H, W = 1080, 1920
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand((1, 3, H, W), device=DEV, generator=g)
    zero = torch.zeros((1, 2, H, W), device=DEV)
    out = softsplat(x, zero, torch.zeros((1, 1, H, W), device=DEV), "soft")
    assert torch.allclose(out, x / (1.0 + 1e-7), rtol=1e-6, atol=0)
    flow = (torch.rand((1, 2, H, W), device=DEV, generator=g) - 0.5) * 4
    flow[:, :, :4] = 0
    flow[:, :, -4:] = 0
    flow[:, :, :, :4] = 0
    flow[:, :, :, -4:] = 0
    s = softsplat(x, flow, None, "sum")
    assert abs(float(s.double().sum()) / float(x.double().sum()) - 1.0) < 1e-5
    # raster identity
    K3, c2w = synth.frame_camera(0, 2, H, W)
    fc = synth.flat_cam(H, W, K3, c2w)
    cam = ops.cam_prep(T(fc))
    ro, rd, uv, _ = ops.get_rays(cam, H, W, 1)
    pts = ro + rd * 2.0
    cloud = torch.cat([pts, torch.rand((H * W, 3), device=DEV, generator=g)], 1)
    r = ops.points_raster(cloud, cloud[:, 3:], cam, 0.4 / (H / 2), 1, H, W, want_fragments=True)
    # pytorch3d pixel centres are at +0.5, rays at integer coordinates: pixel (y,x) sees the
    # point of ray (y,x) iff 0.5^2+0.5^2 < (0.4)^2 is false -> nothing... use the nearest rule:
    # radius 0.4 px < 0.707 px so no pixel is covered
    assert int((r["idx"] >= 0).sum()) == 0
    r = ops.points_raster(cloud, cloud[:, 3:], cam, 0.75 / (H / 2), 1, H, W, want_fragments=True)
    idx = r["idx"][..., 0]
    # every interior pixel is covered by the 4 rays around its centre; the winner is the
    # smallest (z, idx): equal z up to rounding, so just check coverage and the id neighbourhood
    assert float((idx[1:-1, 1:-1] >= 0).float().mean()) == 1.0
    yy, xx = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    own = yy * W + xx
    dpix = (idx - own)[1:-1, 1:-1]
    assert bool(((dpix == 0) | (dpix == 1) | (dpix == W) | (dpix == W + 1)).all())


# ---------------------------------------------------------------- GNT static renderer (A13-A16)
def _gnt_model(golden_dir, depth=2):
    from pgdvs_amd.models.gnt.model import GNTModel

    g = _load(golden_dir, "gnt_small.npz")
    torch.manual_seed(123)
    m = GNTModel(netwidth=64, transformer_depth=depth).eval()
    m.net_coarse.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w_")}, strict=True)
    return m.to(DEV), g


@pytest.mark.parametrize("tag", ["nomask", "dynmask"])
def test_gnt_gather_vs_reference_and_oracle(golden_dir, tag):
    from oracle import gnt_oracle as G

    g = _load(golden_dir, "gnt_small.npz")
    V = int(g["V"])
    cams = ops.cam_prep(T(g["cams_src"]))
    camt = ops.cam_prep(T(g["cam_tgt"]))
    feat_cl = T(g["featmaps"]).permute(0, 2, 3, 1).contiguous()
    out = ops.gnt_gather(T(g["ray_o"]), T(g["ray_d"]), T(g["depth_range"]), int(g["Ss"]), True, camt, cams,
                         T(g["src_rgbs"][0]), feat_cl, T(g["inv_masks"][0, ..., 0]) if tag == "dynmask" else None)
    np.testing.assert_allclose(N(out["pts"]), g["pts"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(N(out["z_vals"]), g["z_vals"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(N(out["mask_inbound"]), g[f"{tag}_mask_inbound"])  # bound tests: exact
    assert np.array_equal(N(out["mask"]), g[f"{tag}_mask"])
    np.testing.assert_allclose(N(out["rgb_feat"]), g[f"{tag}_rgb_feat"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(N(out["ray_diff"]), g[f"{tag}_ray_diff"], rtol=0, atol=5e-5)
    o = G.projector_compute(g["pts"], g["cam_tgt"], g["src_rgbs"][0], g["cams_src"], g["featmaps"],
                            g["inv_masks"][0] if tag == "dynmask" else None)
    np.testing.assert_allclose(N(out["rgb_feat"]), o["rgb_feat"], rtol=0, atol=2e-5)
    assert np.array_equal(N(out["mask"]), o["mask"])
    assert V == out["mask"].shape[2]


@pytest.mark.parametrize("n_rays", [1, 5, 13])
def test_gnt_gather_ragged_sizes(golden_dir, n_rays):
    """item counts that are not multiples of the 8 items a wavefront of the gather kernel owns:
    a prefix of the rays must give exactly the prefix of the full result"""
    g = _load(golden_dir, "gnt_small.npz")
    cams = ops.cam_prep(T(g["cams_src"]))
    camt = ops.cam_prep(T(g["cam_tgt"]))
    feat_cl = T(g["featmaps"]).permute(0, 2, 3, 1).contiguous()
    args = (T(g["depth_range"]), int(g["Ss"]), True, camt, cams, T(g["src_rgbs"][0]), feat_cl, T(g["inv_masks"][0, ..., 0]))
    full = ops.gnt_gather(T(g["ray_o"]), T(g["ray_d"]), *args)
    part = ops.gnt_gather(T(g["ray_o"][:n_rays]), T(g["ray_d"][:n_rays]), *args)
    assert (n_rays * int(g["Ss"]) * int(g["V"])) % 8 != 0 or n_rays == 1
    for k in ("rgb_feat", "ray_diff", "mask", "mask_inbound", "pts", "z_vals"):
        assert torch.equal(part[k], full[k][:n_rays]), k


@pytest.mark.parametrize("tag", ["nomask", "dynmask"])
def test_gnt_forward_vs_reference(golden_dir, tag):
    m, g = _gnt_model(golden_dir)
    with torch.no_grad():
        out, ex = m.net_coarse(T(g[f"{tag}_rgb_feat"]), T(g[f"{tag}_ray_diff"]), T(g[f"{tag}_mask"]), T(g["pts"]), T(g["ray_d"]),
                               ret_view_entropy=True, ret_view_std=True)
    np.testing.assert_allclose(N(out), g[f"{tag}_out"], rtol=0, atol=1e-4)
    for k, v in ex.items():
        np.testing.assert_allclose(N(v), g[f"{tag}_{k}"], rtol=0, atol=1e-4, err_msg=k)


def test_gnt_renderer_end_to_end_vs_reference(golden_dir):
    """feature net (MIOpen) + chunk loop + gather + aggregation + reductions vs BaseRenderer.forward."""
    from pgdvs_amd.models.gnt.renderer import BaseRenderer

    m, g = _gnt_model(golden_dir)
    r = _load(golden_dir, "gnt_render.npz")
    br = BaseRenderer(model_cfg=None)
    br.model = m
    br = br.to(DEV).eval()
    H, W, stride = int(g["H"]), int(g["W"]), int(r["render_stride"])
    camt = ops.cam_prep(T(g["cam_tgt"]))
    ro, rd, uv, shape = ops.get_rays(camt, H, W, stride)
    ray_batch = {"ray_o": ro, "ray_d": rd, "camera": T(g["cam_tgt"][None]), "raw_h": H, "raw_w": W,
                 "depth_range": T(g["depth_range"]), "depth_range_per_ray": False, "src_rgbs": T(g["src_rgbs"]),
                 "src_invalid_masks": T(g["inv_masks"]), "src_cameras": T(g["cams_src"][None])}
    with torch.no_grad():
        ret = br.forward(ray_batch=ray_batch, chunk_size=int(r["chunk_size"]), inv_uniform=True,
                         n_coarse_samples_per_ray=int(g["Ss"]), use_dyn_mask=True, flag_deterministic=True,
                         render_stride=stride, ret_view_entropy=True, ret_view_std=True)
    for k, v in ret["outputs_coarse"].items():
        np.testing.assert_allclose(N(v), r["out_" + k], rtol=0, atol=2e-4, err_msg=k)
    assert ret["outputs_fine"] is None
    # importance re-sampling + second pass (outputs_fine), another chunk size
    with torch.no_grad():
        ret = br.forward(ray_batch=ray_batch, chunk_size=int(r["fine_chunk_size"]), inv_uniform=True,
                         n_coarse_samples_per_ray=int(g["Ss"]), n_fine_samples_per_ray=int(r["n_fine"]), use_dyn_mask=True,
                         flag_deterministic=True, render_stride=stride, ret_view_entropy=True, ret_view_std=True)
    for k, v in ret["outputs_coarse"].items():
        np.testing.assert_allclose(N(v), r["finec_" + k], rtol=0, atol=2e-4, err_msg="coarse " + k)
    assert set(ret["outputs_fine"].keys()) == {k[5:] for k in r if k.startswith("fine_") and k not in ("fine_chunk_size",)}
    for k, v in ret["outputs_fine"].items():
        assert v.shape == r["fine_" + k].shape, k
        np.testing.assert_allclose(N(v), r["fine_" + k], rtol=0, atol=3e-4, err_msg="fine " + k)


def test_pgdvs_renderer_with_gnt_static(golden_dir):
    """PGDVSRenderer with static_renderer=gnt running the network (no rgb_gnt shortcut)."""
    m, g = _gnt_model(golden_dir)
    cfg = load_config(static_renderer="gnt")
    cfg.static_renderer.model_cfg.transformer_depth = 2
    rc = cfg.engine.engine_cfg.render_cfg
    rc.n_coarse_samples_per_ray = int(g["Ss"])
    rc.chunk_size = 500
    rc.gnt_use_masked_spatial_src = False
    rc.gnt_use_dyn_mask = True
    model = PGDVSRenderer(cfg, render_cfg=rc).to(DEV).eval()
    model.static_renderer.model = m
    H, W = int(g["H"]), int(g["W"])
    v = synth.make_video(3, H, W, seed=9)
    d = synth.to_torch(synth.make_view(v, 0, seed=2), DEV)
    d["flat_cam_tgt"] = T(g["cam_tgt"][None])
    d["rgb_src_spatial"] = T(g["src_rgbs"])
    d["dyn_mask_src_spatial"] = T(g["inv_masks"])
    d["flat_cam_src_spatial"] = T(g["cams_src"][None])
    d["depth_range"] = T(g["depth_range"])
    with torch.no_grad():
        ret = model.forward(d, render_cfg=rc)
    for k in ("static_coarse_rgb", "static_coarse_depth", "static_coarse_view_entropy", "static_coarse_view_std",
              "static_coarse_view_std_normalized", "static_coarse_inbound_cnt", "static_coarse_oob_mask",
              "static_coarse_dyn_cnt", "static_coarse_dyn_mask_any", "static_coarse_dyn_mask_all",
              "static_coarse_dyn_mask_thres", "combined_rgb", "render_dyn_rgb"):
        assert k in ret and ret[k].shape[0] == 1 and ret[k].shape[2:] == (H, W), k
    comb = (1 - ret["render_dyn_mask"]) * ret["static_coarse_rgb"] + ret["render_dyn_mask"] * ret["render_dyn_rgb"]
    assert torch.allclose(ret["combined_rgb"], comb, atol=1e-6)
    assert bool(torch.isfinite(ret["combined_rgb"]).all())


@pytest.mark.parametrize("V,want_stats", [(1, True), (4, False), (10, True), (24, True)])
def test_gnt_view_layer_mfma_vs_torch(V, want_stats):
    """The fused fp32-MFMA view-transformer kernel against the plain PyTorch fp32 statement of
    the same layer (random weights, masks with 0/1/all valid views)."""
    import ctypes

    from pgdvs_amd import _lib
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(7 + V)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.2)
    layer = net.view_crosstrans[0]
    R, S = 37, 19  # N = 703: not a multiple of the 32-group tile
    q = torch.randn(R, S, 64, device=DEV)
    feat = torch.randn(R, S, V, 64, device=DEV)
    rd = torch.randn(R, S, V, 4, device=DEV)
    valid = torch.rand(R, S, V, device=DEV) < 0.6
    valid[0, 0] = False
    if V > 1:
        valid[5:9, :, 1] = False  # whole tiles of 16 consecutive (ray,sample) groups without a view: the kernel skips it
    cnt = valid.sum(-1)
    empty = cnt == 0
    valid = valid | empty[..., None]
    cnt = torch.where(empty, torch.full_like(cnt, V), cnt)
    lib = _lib.load()
    buf = ctypes.create_string_buffer(4096)
    lib.pgdvs_prof_report(buf, len(buf))
    lib.pgdvs_prof_enable(1)
    with torch.no_grad():
        out_k, st_k = net._view_layer(layer, q, feat, rd, valid, cnt, want_stats)
    lib.pgdvs_prof_enable(0)
    lib.pgdvs_prof_report(buf, len(buf))
    assert b"gnt_view_layer" in buf.value and b"gnt_ff" in buf.value  # the HIP kernels ran, not the torch path
    ops._GNT_VIEW_ENABLED = False
    try:
        with torch.no_grad():
            out_t, st_t = net._view_layer(layer, q, feat, rd, valid, cnt, want_stats)
    finally:
        ops._GNT_VIEW_ENABLED = True
    np.testing.assert_allclose(N(out_k), N(out_t), rtol=1e-4, atol=2e-5)
    if want_stats:
        for a, b, name in zip(st_k, st_t, ("entropy", "std", "std_norm")):
            np.testing.assert_allclose(N(a), N(b), rtol=1e-3, atol=2e-5, err_msg=name)
    else:
        assert st_k is None


def test_gnt_view_layer_wide_logit_range():
    """view logits tens of units apart: the running softmax has to move its reference logit
    (the rare rescaling branch of the kernel) and still agree with torch's softmax"""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(99)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    layer = net.view_crosstrans[0]
    with torch.no_grad():
        layer.attn.attn_fc[2].weight.mul_(60.0)
        layer.attn.attn_fc[2].bias.add_(torch.randn_like(layer.attn.attn_fc[2].bias) * 5.0)
    R, S, V = 23, 11, 12
    q = torch.randn(R, S, 64, device=DEV)
    feat = torch.randn(R, S, V, 64, device=DEV) * 2.0
    rd = torch.randn(R, S, V, 4, device=DEV)
    valid = torch.rand(R, S, V, device=DEV) < 0.7
    cnt = valid.sum(-1)
    empty = cnt == 0
    valid = valid | empty[..., None]
    cnt = torch.where(empty, torch.full_like(cnt, V), cnt)
    with torch.no_grad():
        out_k, st_k = net._view_layer(layer, q, feat, rd, valid, cnt, True)
        a = layer.attn
        k = a.k_fc(feat)
        logits = a.attn_fc(k - a.q_fc(layer.attn_norm(q))[:, :, None] + a.pos_fc(rd))
        spread = (logits.amax(2) - logits.amin(2)).max().item()
    assert spread > 40.0, spread  # the inputs do exercise the branch
    ops._GNT_VIEW_ENABLED = False
    try:
        with torch.no_grad():
            out_t, st_t = net._view_layer(layer, q, feat, rd, valid, cnt, True)
    finally:
        ops._GNT_VIEW_ENABLED = True
    np.testing.assert_allclose(N(out_k), N(out_t), rtol=2e-4, atol=5e-5)
    for a_, b_, name in zip(st_k, st_t, ("entropy", "std", "std_norm")):
        np.testing.assert_allclose(N(a_), N(b_), rtol=1e-3, atol=5e-5, err_msg=name)


@pytest.mark.parametrize("V", [1, 5, 24])
def test_gnt_embed_mfma_vs_torch(V):
    """rgbfeat_fc + max / std over the views in one MFMA kernel against the torch statements"""
    import ctypes

    from pgdvs_amd import _lib
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(31 + V)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    R, S = 29, 13  # N = 377: not a multiple of the 16-group tile
    x = torch.randn(R, S, V, 35, device=DEV)
    x[..., 3:] += 4.0  # features with a mean far above their spread: the one-pass variance must cope
    lib = _lib.load()
    buf = ctypes.create_string_buffer(4096)
    lib.pgdvs_prof_report(buf, len(buf))
    lib.pgdvs_prof_enable(1)
    with torch.no_grad():
        feat, q0, st = ops.gnt_embed(net.rgbfeat_fc, x, True)
    lib.pgdvs_prof_enable(0)
    lib.pgdvs_prof_report(buf, len(buf))
    assert b"gnt_embed" in buf.value
    with torch.no_grad():
        ref = net.rgbfeat_fc(x)
        s0 = torch.std(ref, dim=2)
        ref_std, ref_stdn = s0.mean(-1), (s0 / (ref.abs().mean(2) + 1e-6)).mean(-1)
    np.testing.assert_allclose(N(feat), N(ref), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(N(q0), N(ref.max(dim=2)[0]), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(N(st[0]), N(ref_std), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(N(st[1]), N(ref_stdn), rtol=1e-4, atol=1e-5)
    feat2, q02, none = ops.gnt_embed(net.rgbfeat_fc, x, False)
    assert none is None and torch.equal(feat2, feat) and torch.equal(q02, q0)


def test_gnt_posfc_mfma_vs_torch():
    """even layers' q_fc(cat(q, posenc(pts), posenc(dir))) split into GEMM parts + the MFMA row
    kernel against the concatenated torch statement"""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT, _posenc

    torch.manual_seed(5)
    net = GNT(netwidth=64, transformer_depth=4).to(DEV).eval()
    R, S = 21, 19
    q = torch.randn(R, S, 64, device=DEV)
    pts = torch.randn(R, S, 3, device=DEV)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=DEV), dim=-1)
    pe_p = _posenc(pts, net.pos_freqs, net.max_log2)
    pe_v = _posenc(dirs, net.view_freqs, net.max_log2)
    with torch.no_grad():
        fused = ops.GntPosFc(net.q_fcs, pe_p, pe_v)
        for i in (0, 2):
            out = fused(i, q)
            ref = net.q_fcs[i](torch.cat((q, pe_p, pe_v[:, None].expand(R, S, -1)), dim=-1))
            np.testing.assert_allclose(N(out), N(ref), rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("S", [1, 37, 256, 300])
def test_gnt_head_vs_torch(S):
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(3 + S)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    with torch.no_grad():
        net.norm.weight.add_(torch.randn_like(net.norm.weight) * 0.3)
        net.norm.bias.add_(torch.randn_like(net.norm.bias) * 0.3)
        q = torch.randn(19, S, 64, device=DEV) * 2 + 0.5
        out = ops.gnt_head(net.norm, net.rgb_fc, q)
        ref = net.rgb_fc(net.norm(q).mean(dim=1))
    np.testing.assert_allclose(N(out), N(ref), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("S", [1, 12, 33, 64, 256])
def test_gnt_ray_layer_mfma_vs_torch(S):
    """Fused ray-transformer kernel (LN, QKV, 4-head attention over the samples of a ray, out_fc,
    residual, FF) and the sample-0 attention row against the PyTorch fp32 statement."""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(100 + S)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.2)
    layer = net.view_selftrans[0]
    R = 19
    q = torch.randn(R, S, 64, device=DEV) * 1.5
    with torch.no_grad():
        out_k, w_k = GNT._ray_layer(layer, q, True)
    ops._GNT_VIEW_ENABLED = False
    try:
        with torch.no_grad():
            out_t, w_t = GNT._ray_layer(layer, q, True)
    finally:
        ops._GNT_VIEW_ENABLED = True
    np.testing.assert_allclose(N(out_k), N(out_t), rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(N(w_k), N(w_t), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(N(w_k).sum(1), 1.0, rtol=1e-5)
