"""CPU: the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Float rows: tolerance stated per test.  Integer /
index rows: bit-exact."""
import numpy as np
import pytest

from oracle import oracle as orc


def _load(golden_dir, name):
    return dict(np.load(golden_dir / name))


@pytest.mark.parametrize("name", ["rays_a.npz", "rays_b.npz"])
def test_rays(golden_dir, name):
    g = _load(golden_dir, name)
    ro, rd, uv, shape = orc.get_batched_rays(g["flat_cam"], int(g["H"]), int(g["W"]), int(g["stride"]))
    assert tuple(shape) == tuple(g["render_hw"])
    assert np.array_equal(uv, g["uvs"])  # integer pixel centres: exact
    assert np.array_equal(ro, g["rays_o"])
    np.testing.assert_allclose(rd, g["rays_d"], rtol=2e-6, atol=2e-6)


def test_project(golden_dir):
    g = _load(golden_dir, "project.npz")
    uv = orc.project(g["flat_cam"], g["xyz"])
    # points behind the camera are clamped to +-1e6 by both
    np.testing.assert_allclose(uv, g["uv"], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("case", range(8))
def test_compute_dyn_pcl(golden_dir, case):
    g = _load(golden_dir, f"dyn_pcl_{case}.npz")
    r = orc.compute_dyn_pcl(
        dyn_mask_1=g["dyn_mask_1"], rgb_1=g["rgb_1"], depth_1=g["depth_1"], flow_12=g["flow_12"],
        flow_12_occ_mask=g["flow_12_occ_mask"], rgb_2=g["rgb_2"], depth_2=g["depth_2"],
        flat_cam_1=g["flat_cam_1"], flat_cam_2=g["flat_cam_2"], flat_cam_tgt=g["flat_cam_tgt"],
        time_1=float(g["time_1"]), time_2=float(g["time_2"]), time_tgt=float(g["time_tgt"]),
        dyn_render_use_flow_consistency=bool(g["use_flow_consistency"]),
        dyn_pcl_remove_outlier=bool(g["remove_outlier"]), dyn_pcl_outlier_knn=int(g["outlier_knn"]),
        dyn_pcl_outlier_std_thres=float(g["outlier_std_thres"]),
    )
    # integer path: which pixels survive (mask compaction + bounds + outlier flags)
    assert np.array_equal(r["valid_dyn_mask_1"], g["out_valid_dyn_mask_1"])
    assert r["pcl"].shape == g["out_pcl"].shape
    np.testing.assert_allclose(r["pcl"], g["out_pcl"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["pcl_rgbs"], g["out_pcl_rgbs"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["pcl_nn_dist_thres"], g["out_nn_dist_thres"], rtol=1e-4)
    np.testing.assert_allclose(r["flow_1_to_tgt"], g["out_flow_1_to_tgt"], rtol=1e-4, atol=2e-4)


def test_backwarp_l1(golden_dir):
    g = _load(golden_dir, "backwarp_l1.npz")
    l1 = orc.backwarp_l1(g["rgb1"][0], g["rgb2"][0], g["flow"][0])
    np.testing.assert_allclose(l1, g["l1"][0, 0], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("mode", ["sum", "avg", "linear", "soft", "soft-zeroeps", "soft-clipeps"])
def test_softsplat_modes(golden_dir, mode):
    g = _load(golden_dir, "softsplat_modes.npz")
    metric = None if mode in ("sum", "avg") else (g["ten_metric"] if mode != "linear" else np.abs(g["ten_metric"]) + 0.1)
    out = orc.softsplat(g["ten_in"], g["ten_flow"], metric, mode)
    ref = g["out_" + mode.replace("-", "_")]
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["forward_a.npz", "forward_b.npz"])
def test_forward(golden_dir, name):
    g = _load(golden_dir, name)
    data = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    cfg = dict(
        dyn_render_use_flow_consistency=bool(g["use_flow_consistency"]), dyn_pcl_remove_outlier=bool(g["remove_outlier"]),
        dyn_pcl_outlier_knn=int(g["outlier_knn"]), dyn_pcl_outlier_std_thres=float(g["outlier_std_thres"]),
        dyn_render_type="softsplat",
    )
    ret = orc.render_view(data, cfg, static_noise=g["static_noise"], alpha=100.0)
    assert np.array_equal(ret["render_dyn_mask"], g["out_render_dyn_mask"])  # thresholded mask: exact
    for k in ["render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]:
        np.testing.assert_allclose(ret[k], g["out_" + k], rtol=0, atol=1e-4, err_msg=k)


def test_static_aggregation(golden_dir):
    g = _load(golden_dir, "static_agg.npz")
    S, H, W = g["depths"].shape
    K3s = np.stack([orc.hwf_to_K(*g["hwf"][i]) for i in range(S)])
    pcl1 = orc.compute_pcl(H, W, K3s[1], g["c2ws"][1], g["depths"][1])
    np.testing.assert_allclose(pcl1, g["pcl_frame1"], rtol=1e-5, atol=1e-6)
    # occupancy bitmap: integer path, exact (fed with the reference's own points)
    pm = orc.static_proj_mask(g["pcl_frame1"], K3s[2], np.linalg.inv(g["c2ws"][2]), H, W)
    assert np.array_equal(pm, g["proj_mask_1_into_2"])
    rgbs = g["imgs"].astype(np.float32) / 255.0
    st = orc.aggregate_static_pcl(rgbs, g["depths"], g["dyn_masks"], K3s, g["c2ws"])
    assert st.shape == g["st_pcl_rgb"].shape  # same point set size => same occupancy decisions
    np.testing.assert_allclose(st[:, :3], g["st_pcl_rgb"][:, :3], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[:, 3:], g["st_pcl_rgb"][:, 3:], rtol=0, atol=1e-6)


# ---------------------------------------------------------------- GNT rows (A13-A15)
@pytest.mark.parametrize("tag", ["nomask", "dynmask"])
def test_gnt_oracle_vs_reference(golden_dir, tag):
    from oracle import gnt_oracle as G

    g = _load(golden_dir, "gnt_small.npz")
    Wt = {k[2:]: v for k, v in g.items() if k.startswith("w_")}
    R = g["ray_o"].shape[0]
    pts, z = G.sample_along_camera_ray(g["ray_o"], g["ray_d"], np.broadcast_to(g["depth_range"], (R, 2)), int(g["Ss"]))
    np.testing.assert_allclose(pts, g["pts"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(z, g["z_vals"], rtol=1e-6, atol=1e-6)
    pr = G.projector_compute(g["pts"], g["cam_tgt"], g["src_rgbs"][0], g["cams_src"], g["featmaps"],
                             g["inv_masks"][0] if tag == "dynmask" else None)
    assert np.array_equal(pr["mask_inbound"], g[f"{tag}_mask_inbound"])  # in-bounds / in-front tests: exact
    assert np.array_equal(pr["mask"], g[f"{tag}_mask"])
    np.testing.assert_allclose(pr["rgb_feat"], g[f"{tag}_rgb_feat"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pr["ray_diff"], g[f"{tag}_ray_diff"], rtol=0, atol=5e-5)
    out, ex = G.gnt_forward(Wt, g[f"{tag}_rgb_feat"], g[f"{tag}_ray_diff"], g[f"{tag}_mask"], g["pts"], g["ray_d"])
    np.testing.assert_allclose(out, g[f"{tag}_out"], rtol=0, atol=1e-4)
    for k in ex:
        np.testing.assert_allclose(ex[k], g[f"{tag}_{k}"], rtol=0, atol=1e-4, err_msg=k)


def _depth8(golden_dir):
    """inputs + weights of gnt_depth8.npz from their seeded generators, checked against the fixture's checksums"""
    import sys

    sys.path.insert(0, str(golden_dir))
    import gnt_depth8_inputs as GI

    g = _load(golden_dir, "gnt_depth8.npz")
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    shapes = {k: tuple(v.shape) for k, v in GNT(netwidth=64, transformer_depth=8).state_dict().items()}
    assert list(shapes) == [str(k) for k in g["state_dict_keys"]]  # the mirror's parameter names ARE the reference's
    w = GI.make_weights(shapes)
    assert abs(GI.checksum(w) - float(g["weights_checksum"])) <= 1e-9 * abs(float(g["weights_checksum"])) + 1e-9
    assert sum(v.size for v in w.values()) == int(g["n_params"])
    return GI, g, w


@pytest.mark.parametrize("case", ["v10", "v24"])
def test_gnt_oracle_vs_reference_at_depth_8(golden_dir, case):
    """the reference's GNT.forward at the depth it is configured with (8 layers, 256 samples, 10 / 24 views;
    configs/static_renderer/gnt.yaml:9, transformer_network.py:423-539): the oracle on a subset of rays (numpy; rays are
    independent), incl. the rays without / with one / with all views valid"""
    from oracle import gnt_oracle as G

    GI, g, w = _depth8(golden_dir)
    x = GI.make_inputs(case)
    assert abs(GI.checksum(x) - float(g[f"{case}_inputs_checksum"])) <= 1e-9 * abs(float(g[f"{case}_inputs_checksum"])) + 1e-9
    sel = np.array([0, 1, 2, 5])
    out, ex = G.gnt_forward(w, x["rgb_feat"][sel], x["ray_diff"][sel], x["mask"][sel], x["pts"][sel], x["ray_d"][sel])
    np.testing.assert_allclose(out[:, :3], g[f"{case}_out"][sel, :3], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out[:, 3:], g[f"{case}_out"][sel, 3:], rtol=1e-3, atol=1e-6)  # sample weights (~1/256)
    for k in ex:
        np.testing.assert_allclose(ex[k], g[f"{case}_{k}"][sel], rtol=0, atol=1e-4, err_msg=k)


# ---------------------------------------------------------------- A17 tracker-window aggregation
def _track_case(g, c):
    wb = bool(g[f"c{c}_with_base"])
    th = g[f"c{c}_base_thres"]
    rc = dict(dyn_pcl_outlier_knn=int(g[f"c{c}_knn"]), dyn_pcl_track_track2base_thres_mult=50, dyn_pcl_outlier_std_thres=0.1)
    return rc, (g[f"c{c}_base_pts"] if wb else None), (g[f"c{c}_base_rgb"] if wb else None), (None if np.isnan(th) else th)


def test_track_prepare_data(golden_dir):
    g = _load(golden_dir, "track_pcl.npz")
    dft = orc.track_prepare_data({k[5:]: v for k, v in g.items() if k.startswith("data_")}, 0)
    assert np.array_equal(dft["times"], g["dfk_times"]) and dft["time_tgt"] == g["dfk_time_tgt"][0]
    assert np.array_equal(np.nonzero(dft["kind"] == 1)[0], g["dfk_idx_closest"])
    assert np.array_equal(np.nonzero(dft["kind"] == 2)[0], g["dfk_idx_real"])
    for a, b in (("rgbs", "dfk_rgbs"), ("depths", "dfk_depths"), ("flat_cams", "dfk_cams")):
        assert np.array_equal(dft[a], g[b])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_track_compute_pcl_for_tgt(golden_dir, case):
    g = _load(golden_dir, "track_pcl.npz")
    dft = orc.track_prepare_data({k[5:]: v for k, v in g.items() if k.startswith("data_")}, 0)
    rc, bp, br, th = _track_case(g, case)
    pcl, rgb, _ = orc.track_compute_pcl_for_tgt(dft, g[f"c{case}_tracks"], g[f"c{case}_vis"], rc, bp, br, th)
    # same point-set size => same validity / frame-pair / filter decisions
    assert pcl.shape == g[f"c{case}_out_pcl"].shape
    np.testing.assert_allclose(pcl, g[f"c{case}_out_pcl"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgb, g[f"c{case}_out_rgb"], rtol=0, atol=1e-6)


# ---------------------------------------------------------------- softsplat backward (8f-4)
def test_softsplat_backward_vs_autograd_golden(golden_dir):
    """oracle restatement of softsplat_ingrad / softsplat_flowgrad vs torch autograd through the
    restated forward (tests/golden/make_golden_softsplat_bwd.py); tolerance: fp32 sums of <= 12
    products with |values| up to ~10"""
    g = _load(golden_dir, "softsplat_bwd.npz")
    gi, gf = orc.softsplat_bwd_raw(g["ten_in"], g["ten_flow"], g["grad_out"])
    np.testing.assert_allclose(gi, g["sum_grad_in"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(gf, g["sum_grad_flow"], rtol=0, atol=1e-5)
    assert np.all(gi[0, :, 0, :4] == 0) and np.all(gf[0, :, 0, :4] == 0)  # targets far outside the image


# ---------------------------------------------------------------- A9: point-major evaluation == naive loop
@pytest.mark.parametrize("H,W,K,radius", [(40, 64, 3, 0.05), (64, 40, 1, 0.03), (48, 48, 8, 0.08), (33, 57, 3, 0.2)])
def test_pointmajor_raster_equals_naive(H, W, K, radius):
    """orc_raster_points_pointmajor (the full-frame checker of the 1080p tests) returns the naive O(pixels x points)
    loop's idx / zbuf / dist2 bit for bit: random clouds with exact z ties, points behind the camera, NaN / inf
    coordinates, discs grazing pixel centres and points far outside the frame."""
    rng = np.random.default_rng(H * 1000 + W)
    N = 6000
    ax, ay = (W / H, 1.0) if W > H else (1.0, H / W)
    ndc = np.empty((N, 3), np.float32)
    ndc[:, 0] = rng.uniform(-1.3 * ax, 1.3 * ax, N)
    ndc[:, 1] = rng.uniform(-1.3 * ay, 1.3 * ay, N)
    ndc[:, 2] = rng.uniform(0.5, 4.0, N)
    ndc[rng.choice(N, 800, replace=False), 2] = np.float32(1.25)  # exact ties: the order among them is the index order
    ndc[rng.choice(N, 200, replace=False), 2] = -0.5               # behind the camera
    ndc[rng.choice(N, 20, replace=False), 2] = 0.0
    bad = rng.choice(N, 30, replace=False)
    ndc[bad[:10], 0] = np.nan
    ndc[bad[10:20], 1] = np.inf
    ndc[bad[20:], 2] = np.nan
    # discs whose edge passes (nearly) through a pixel centre: centre + radius along x, nudged by a few ulp
    xs = np.array([-(W / min(H, W)) + (2 * (W - 1 - x) + 1) / min(H, W) for x in range(W)], np.float32)
    for j, i in enumerate(rng.choice(N, 300, replace=False)):
        ndc[i, 0] = np.nextafter(np.float32(xs[j % W] + np.float32(radius)), np.float32(10 * (j % 3 - 1)))
    ref = orc.rasterize_points_window(ndc, H, W, radius, K, 0, H, 0, W)
    got = orc.rasterize_points_pointmajor(ndc, H, W, radius, K)
    assert np.array_equal(ref[0], got[0])
    assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
    assert np.array_equal(ref[2].view(np.uint32), got[2].view(np.uint32))
    assert (ref[0] >= 0).mean() > 0.3  # the case is not empty
