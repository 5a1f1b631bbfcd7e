"""GPU tests added in round 3 (MI355X): the randomised GPU-vs-oracle sweep over the kernels with data-dependent
fast paths (formerly the uncollected tests/fuzz_parity.py), the splat kernel's own noise, the rasteriser's
right-sized workspace with its overflow status, and the bounded receive ring of the image gather."""
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


def _renderer(static="geo", **over):
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

    cfg = load_config(static_renderer=static)
    rc = cfg.engine.engine_cfg.render_cfg
    for k, v in over.items():
        rc[k] = v
    return PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(DEV).eval(), rc


# ---------------------------------------------------------------- randomised sweep (fixed budget: 8 seeds)
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_knn_raster_view_vs_oracle(seed):
    """many seeds, ragged sizes: kNN (thread-per-query pass and ring search; surface + clusters + duplicates; K inside
    and outside the thread-per-query set), rasteriser (layered depths with exact ties, random radius / K) and a whole
    small view (splat with the flag pre-pass, outlier filter on) -- bit-exact / 1e-4 against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(60, 6000))
    K = int(rng.choice([4, 8, 16, 20, 50, 7, 30]))
    u = rng.uniform(-1, 1, (n, 2))
    pts = np.stack([u[:, 0], u[:, 1], 2 + 0.3 * np.sin(3 * u[:, 0]) + rng.normal(0, 0.003, n)], 1)
    pts[: n // 20] += rng.normal(0, 0.7, (n // 20, 3))
    pts[n // 2: n // 2 + n // 30] = pts[:n // 30]
    pts = pts.astype(np.float32)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    a = N(ops.knn_mean_dist(T(pts), cnt, K, algo=2))[:n]
    assert np.array_equal(a.view(np.uint32), orc.knn_mean_dist(pts, K).view(np.uint32)), f"knn n={n} K={K}"

    H, W = int(rng.integers(20, 70)), int(rng.integers(20, 70))
    m, Kp, radius = int(rng.integers(500, 20000)), int(rng.integers(1, 9)), float(rng.uniform(0.01, 0.09))
    fc = synth.flat_cam(H, W, *synth.frame_camera(1, 4, H, W))
    z = 1.0 + 0.2 * rng.integers(0, 5, m) + np.where(rng.random(m) < 0.4, 0.0, rng.uniform(0, 0.05, m))
    xy = rng.uniform(-1.3, 1.3, (m, 2)) * z[:, None] * 0.6
    p3 = np.concatenate([xy, z[:, None]], 1).astype(np.float32)
    cloud = T(np.concatenate([p3, rng.random((m, 3), dtype=np.float32)], 1))
    r = ops.points_raster(cloud, cloud[:, 3:], ops.cam_prep(T(fc)), radius, Kp, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(p3, fc, H, W, radius, Kp)
    assert np.array_equal(N(r["idx"]), idx), f"raster {H}x{W} n={m} K={Kp} r={radius:.3f}"
    assert np.array_equal(N(r["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(r["dist2"]).view(np.uint32), d2.view(np.uint32))

    Hh, Ww, S = int(rng.integers(40, 90)), int(rng.integers(48, 120)), 3
    v = synth.make_video(S, Hh, Ww, seed=seed)
    d = synth.make_view(v, int(rng.integers(0, 2)), seed=seed)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=20, st_render_pcl_pts_per_pixel=3)
    cloud_s = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"] = T(cloud_s.astype(np.float32))[None]
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    od = dict(d)
    od["st_pcl_rgb"] = cloud_s[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    np.testing.assert_allclose(N(ret["combined_rgb"]), o["combined_rgb"], rtol=0, atol=1e-4)


# ---------------------------------------------------------------- the splat kernel's own noise (A8, :177-182)
def test_splat_noise_field_is_standard_normal_and_advances():
    H, W = 270, 480
    st = ops.splat_rng_state(DEV, seed=1234)
    f0 = N(ops.splat_noise_field(st, H, W)).astype(np.float64)
    assert f0.shape == (3, H, W) and np.isfinite(f0).all()
    n = f0.size
    assert abs(f0.mean()) < 4 / np.sqrt(n) and abs(f0.std() - 1) < 0.01
    assert abs((f0 ** 3).mean()) < 0.02 and abs((f0 ** 4).mean() - 3) < 0.06
    # channels and neighbouring pixels are uncorrelated
    assert abs(np.corrcoef(f0[0].ravel(), f0[1].ravel())[0, 1]) < 0.01 and abs(np.corrcoef(f0[0].ravel(), f0[2].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(f0[0, :, 1:].ravel(), f0[0, :, :-1].ravel())[0, 1]) < 0.01
    # E clamp(X, 0, 1) = phi(0) - phi(1) + 1 - Phi(1)
    assert abs(np.clip(f0, 0, 1).mean() - 0.315626) < 2e-3
    # same state -> same field; another seed or another draw number -> another field
    assert np.array_equal(N(ops.splat_noise_field(ops.splat_rng_state(DEV, seed=1234), H, W)), f0.astype(np.float32))
    assert not np.array_equal(N(ops.splat_noise_field(ops.splat_rng_state(DEV, seed=1235), H, W)), f0.astype(np.float32))
    st[1] += 1
    assert abs(np.corrcoef(N(ops.splat_noise_field(st, H, W)).ravel(), f0.ravel())[0, 1]) < 0.01


def test_renderer_draws_its_noise_in_the_splat_kernel():
    """no ``static_noise`` in the data dict: the reference draws torch.randn_like per forward (:181); here the scatter
    kernel draws the field for the pixels that consume it.  The images equal those of the injected-noise path (and of
    the oracle) fed with the field of that draw, and every forward uses a new draw."""
    H, W, S = 96, 160, 3
    v = synth.make_video(S, H, W, seed=11)
    d = synth.make_view(v, 1, seed=3)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=16, st_render_pcl_pts_per_pixel=3)
    cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"] = T(cloud)[None]
    data.pop("static_noise")
    with torch.no_grad():
        r0 = model.forward(dict(data), render_cfg=rc)
        r1 = model.forward(dict(data), render_cfg=rc)
    state = model.dyn_renderer.splat_rng_state(data["rgb_src_temporal"].device)  # (the current stream's state)
    assert int(state[1]) == 2  # two forwards, two draws
    for draw, ret in ((0, r0), (1, r1)):
        st = state.clone()
        st[1] = draw
        field = ops.splat_noise_field(st, H, W)
        inj = dict(data)
        inj["static_noise"] = field[None]
        with torch.no_grad():
            ri = model.forward(inj, render_cfg=rc)
        assert torch.equal(ret["render_dyn_mask"], ri["render_dyn_mask"])
        assert torch.allclose(ret["combined_rgb"], ri["combined_rgb"], rtol=0, atol=1e-6)  # (float atomics: to rounding)
        od = dict(d)
        od["st_pcl_rgb"] = cloud[None]
        o = orc.render_view(od, dict(rc), static_noise=N(field)[None], alpha=100.0)
        assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
        np.testing.assert_allclose(N(ret["combined_rgb"]), o["combined_rgb"], rtol=0, atol=1e-4)
    # the noise shows (only) where static source pixels splat beside dynamic content
    diff = (r0["render_dyn_rgb"] - r1["render_dyn_rgb"]).abs().amax(1)[0]
    assert float(diff.max()) > 1e-3 and float((diff > 0).float().mean()) < 0.2


# ---------------------------------------------------------------- rasteriser workspace sized by a row bound
def _cloud_case(n=4000, cap=20000, H=72, W=96, seed=0):
    rng = np.random.default_rng(seed)
    pts = np.zeros((cap, 6), np.float32)
    pts[:n] = np.concatenate([rng.uniform(-0.6, 0.6, (n, 2)), rng.uniform(1.0, 2.0, (n, 1)), rng.random((n, 3))], 1)
    pts[n:] = np.nan  # rows beyond the count must never be read into the image
    cam = ops.cam_prep(T(synth.flat_cam(H, W, np.array([[70.0, 0, W / 2], [0, 70.0, H / 2], [0, 0, 1]]), np.eye(4)).astype(np.float32)))
    return T(pts), cam, H, W


def test_points_raster_row_bound_matches_capacity_sized_call():
    from pgdvs_amd import _lib

    pts, cam, H, W = _cloud_case()
    cnt = torch.tensor([4000], dtype=torch.int64, device=DEV)
    full = ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, want_fragments=True)
    bnd = ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, want_fragments=True, row_bound=4500)
    for k in ("idx", "zbuf", "dist2", "rgb", "mask"):
        assert torch.equal(full[k], bnd[k]), k
    assert int(bnd["status"]) == 0
    ops.check_raster_status(bnd["status"])
    lib = _lib.load()
    # the workspace follows the row bound (its tile lists); the segments of the direct binning pass are only part of workspaces
    # sized for sparse clouds (below 2.2 rows per pixel: round 6), so the proportion is checked between two sizes without them
    ws = lib.pgdvs_points_raster_workspace_bytes
    assert ws(4500, H, W, 0.03) < ws(20000, H, W, 0.03) and ws(20000, H, W, 0.03) < ws(100000, H, W, 0.03) / 3
    # exactly at the bound: fine
    assert int(ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, row_bound=4000)["status"]) == 0


def test_points_raster_row_bound_overflow_and_negative_count_are_reported():
    from pgdvs_amd._lib import PgdvsHipError

    pts, cam, H, W = _cloud_case()
    cnt = torch.tensor([4000], dtype=torch.int64, device=DEV)
    cut = ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, want_fragments=True, row_bound=3000)
    assert int(cut["status"]) == 1
    with pytest.raises(PgdvsHipError, match="row bound"):
        ops.check_raster_status(cut["status"])
    # what was drawn is the first 3000 rows, nothing from beyond
    ref = ops.points_raster(pts[:3000], pts[:3000, 3:], cam, 0.03, 3, H, W, want_fragments=True)
    assert torch.equal(cut["idx"], ref["idx"]) and torch.equal(cut["rgb"], ref["rgb"])
    neg = ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=torch.tensor([-1], dtype=torch.int64, device=DEV),
                            want_fragments=True, row_bound=3000)
    assert int(neg["status"]) == 2 and int((neg["idx"] >= 0).sum()) == 0
    with pytest.raises(PgdvsHipError, match="negative"):
        ops.check_raster_status(neg["status"])
    zero = ops.points_raster(pts, pts[:, 3:], cam, 0.03, 3, H, W, n_points_dev=cnt, row_bound=0)
    assert int(zero["status"]) == 1 and float(zero["mask"].sum()) == 0.0


def test_eval_step_raises_on_a_cut_or_failed_static_cloud():
    """harness.eval_step (which synchronises for its metrics anyway) reads the geometry path's status words: a cloud
    that outgrew the rasteriser's row bound, or an aggregation that reported an error (count -1), must not pass as a
    truncated / blank static image"""
    from pgdvs_amd import harness
    from pgdvs_amd._lib import PgdvsHipError

    H, W, S = 64, 96, 3
    v = synth.make_video(S, H, W, seed=2)
    d = synth.make_view(v, 0, seed=1)
    model, rc = _renderer("geo", st_render_pcl_pts_per_pixel=3)
    cloud, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=S * H * W)
    n = ops.checked_count(cnt, "agg")
    hd = {k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in d.items()}
    hd["st_pcl_rgb"], hd["st_pcl_rgb_count"] = cloud[None], cnt
    hd["rgb_tgt"] = torch.rand(1, H, W, 3)
    hd["eval_mask"] = torch.zeros(1, H, W, 3)
    hd["st_pcl_rgb_row_bound"] = n + 100
    m_ok, ex = harness.eval_step(model, hd, rc, device=DEV, return_images=True)
    assert int(ex["ret"]["geo_static_raster_status"]) == 0
    ref = harness.eval_step(model, {k: x for k, x in hd.items() if k != "st_pcl_rgb_row_bound"}, rc, device=DEV)
    assert float(m_ok["eval/psnr_full_combined"]) == float(ref["eval/psnr_full_combined"])
    hd["st_pcl_rgb_row_bound"] = n - 1
    with pytest.raises(PgdvsHipError, match="row bound"):
        harness.eval_step(model, hd, rc, device=DEV)
    hd["st_pcl_rgb_row_bound"] = n + 100
    hd["st_pcl_rgb_count"] = torch.tensor([-1], dtype=torch.int64, device=DEV)
    with pytest.raises(PgdvsHipError, match="error flag"):
        harness.eval_step(model, hd, rc, device=DEV)


def test_no_tracker_zero_outputs_cannot_be_written_through():
    H, W, S = 48, 64, 3
    v = synth.make_video(S, H, W, seed=2)
    model, rc = _renderer("geo")
    cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    data = synth.to_torch(synth.make_view(v, 0, seed=1), DEV)
    data["st_pcl_rgb"] = T(cloud)[None]
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    z = ret["render_dyn_temporal_track_rgb"]
    assert z.shape == ret["render_dyn_rgb"].shape and float(z.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        z.add_(1.0)
    with torch.no_grad():
        again = model.forward(data, render_cfg=rc)
    assert float(again["render_dyn_temporal_track_mask"].abs().sum()) == 0.0


def test_splat_composite_converging_and_diverging_flows():
    """pgdvs_dyn_splat_composite directly against the oracle's softsplat_img (pgdvs_renderer_base.py:59-89,
    pgdvs_renderer_dyn.py:177-202) on flows that stress the scatter's LDS accumulation: a 32 x 32 block whose pixels all
    land on the SAME four target pixels (every lane of a tile races for the same accumulators: the compare-and-swap add
    must lose nothing), a block that spreads to three times its size (corners outside a tile's 40 x 40 window: global
    atomics), static pixels with noise next to both.  fp32 sums in another order: 1e-4."""
    H, W = 64, 96
    rng = np.random.default_rng(7)
    rgb1 = rng.random((H, W, 3), dtype=np.float32)
    rgb2 = rng.random((H, W, 3), dtype=np.float32)
    flow12 = (rng.standard_normal((H, W, 2)) * 0.7).astype(np.float32)
    noise = rng.standard_normal((3, H, W)).astype(np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    f1t = np.zeros((2, H, W), np.float32)
    f1t[0] = 0.25 * np.sin(ys / 5.0)  # the static pixels drift a little
    f1t[1] = 0.25 * np.cos(xs / 7.0)
    mask = np.zeros((H, W), np.float32)
    a = (slice(0, 32), slice(0, 32))  # converging block -> target (40.3, 20.6)
    f1t[0][a] = 40.3 - xs[a]
    f1t[1][a] = 20.6 - ys[a]
    mask[a] = 1.0
    b = (slice(32, 48), slice(48, 64))  # diverging block: x3 about its corner
    f1t[0][b] = 2.0 * (xs[b] - 48.0) + 0.4
    f1t[1][b] = 2.0 * (ys[b] - 32.0) * 0.5 + 0.2
    mask[b] = 1.0
    static_rgb = rng.random((3, H, W), dtype=np.float32)
    alpha = 100.0

    m4 = mask[None, None]
    c1 = np.transpose(rgb1, (2, 0, 1))[None] * m4 + np.clip(noise, 0.0, 1.0)[None] * (1.0 - m4)
    c2 = np.transpose(rgb2, (2, 0, 1))[None]
    f12 = np.transpose(flow12, (2, 0, 1))[None]
    splat_full, metric = orc.softsplat_img(c1.astype(np.float32), f1t[None], c2, f12, alpha)
    splat_mask, _ = orc.softsplat_img(m4.astype(np.float32), f1t[None], c2, f12, alpha, metric=metric)
    want_mask = (splat_mask > 1e-3).astype(np.float32)[0, 0]
    want_rgb = (splat_full * want_mask[None, None])[0]
    # pixels whose splatted mask sits on the 1e-3 threshold may flip with the summation order
    sure = np.abs(splat_mask[0, 0] - 1e-3) > 1e-5

    got_rgb, got_mask, comb, comb_st, comb_dy = ops.dyn_splat_composite(
        T(rgb1), T(rgb2), T(flow12), T(f1t), T(mask), T(noise), alpha, T(static_rgb))
    torch.cuda.synchronize()
    got_rgb, got_mask = N(got_rgb), N(got_mask)
    assert want_mask.sum() > 200 and (want_mask[18:24, 38:44] > 0).any()
    np.testing.assert_array_equal(got_mask[sure], want_mask[sure])
    np.testing.assert_allclose(got_rgb[:, sure], want_rgb[:, sure], rtol=1e-4, atol=1e-4)
    want_comb = (1.0 - want_mask)[None] * static_rgb + want_mask[None] * want_rgb
    np.testing.assert_allclose(N(comb)[:, sure], want_comb[:, sure], rtol=1e-4, atol=1e-4)
    # the same call again: the LDS adds lose nothing under contention, so the two results agree to rounding
    again = N(ops.dyn_splat_composite(T(rgb1), T(rgb2), T(flow12), T(f1t), T(mask), T(noise), alpha, T(static_rgb))[0])
    np.testing.assert_allclose(again[:, sure], got_rgb[:, sure], rtol=1e-5, atol=1e-5)


def test_static_aggregate_adversarial_cameras():
    """A12 on scenes built to strain the fp32 screening's error bound (tests/agg_stress.py: large rotations, depths over
    four decades, projections far outside / behind / next to the image plane, integer-valued projections of planes):
    the cloud equals the oracle's fp64 evaluation (nvidia_eval_pure_geo.py:257-277, 407-451) bit for bit."""
    import importlib.util
    import pathlib

    spec = importlib.util.spec_from_file_location(
        "agg_stress", pathlib.Path(__file__).resolve().parent / "agg_stress.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed in range(40):
        rgbs, depths, masks, K3s, c2ws = mod.scene(seed)
        want = orc.aggregate_static_pcl(rgbs, depths, masks, K3s, c2ws).astype(np.float32)
        cloud, cnt = ops.static_aggregate(T(rgbs), T(depths), T(masks), K3s, c2ws)
        k = ops.checked_count(cnt, "pgdvs_static_aggregate")
        got = N(cloud[:k])
        assert got.shape == want.shape, f"seed {seed}: {k} points, the oracle has {want.shape[0]}"
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"seed {seed}"
