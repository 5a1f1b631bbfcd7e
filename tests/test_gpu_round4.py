"""GPU tests added in round 4 (MI355X): the one-native-call-per-view entry point against the per-op path and the oracle,
the benchmark's own arrangement (resident video, cloud aggregated inside the call, bounded buffers, views in flight,
noise drawn in the kernel) at BASELINE.json's configs[2] against the oracle, the evaluator's metric kernel against the
torch statement pinned by the reference's fixture, scenes with other statistics than the nominal one, and the
counters of the fast paths' exits."""
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


def _video_dict(v, capacity=None):
    d = {"rgbs": T(v["rgbs"]), "depths": T(v["depths"]), "dyn_masks": T(v["dyn_masks"]).view(torch.uint8), "K3s": v["K3s"],
         "c2ws": v["c2ws"]}
    if capacity is not None:
        d["capacity"] = capacity
    return d


IMAGE_KEYS = ["geo_static_rgb", "render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]


# ---------------------------------------------------------------- one native call per view
@pytest.mark.parametrize("outlier,use_xyz,use_count,side", [(True, False, False, False), (True, True, True, False),
                                                            (False, True, True, True), (True, False, True, True)])
def test_native_view_call_equals_per_op_path(outlier, use_xyz, use_count, side, monkeypatch):
    """PGDVSRenderer.forward through pgdvs_view_geo_forward (ONE C-ABI call) and through the ~85 per-op calls: the same
    kernels on the same inputs -- static image, masks and status identical, splat images to float-atomic rounding; and
    both against the oracle"""
    H, W, S = 120, 200, 5
    v = synth.make_video(S, H, W, seed=31)
    d = synth.make_view(v, 2, frac=0.3, seed=4)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=outlier, dyn_pcl_outlier_knn=20, st_render_pcl_pts_per_pixel=3,
                          st_render_pcl_pt_radius=0.015)
    cap = S * H * W
    cloud, cnt, xyz = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=cap,
                                           return_xyz=True)
    n = ops.checked_count(cnt, "agg")
    data = synth.to_torch(d, DEV)
    if use_count:
        data["st_pcl_rgb"], data["st_pcl_rgb_count"] = cloud[None], cnt
        data["st_pcl_rgb_row_bound"] = n + 1000
        if use_xyz:
            data["st_pcl_xyz"] = xyz[None]
    else:
        data["st_pcl_rgb"] = cloud[None, :n].contiguous()
    # (True: the caller's own second stream; False: none -- everything on the caller's stream; a caller that names neither gets
    # the renderer's default second stream: test_gpu_round6.py)
    data["_side_stream"] = torch.cuda.Stream(device=DEV) if side else False
    assert model._native_view_ok(data, rc)
    with torch.no_grad():
        rn = model.forward(dict(data), render_cfg=rc)
        monkeypatch.setenv("PGDVS_NATIVE_VIEW", "0")
        assert not model._native_view_ok(data, rc)
        rp = model.forward(dict(data), render_cfg=rc)
    torch.cuda.synchronize()
    assert set(rp) <= set(rn), set(rp) - set(rn)
    assert torch.equal(rn["geo_static_rgb"], rp["geo_static_rgb"]) and torch.equal(rn["geo_static_mask"], rp["geo_static_mask"])
    assert torch.equal(rn["render_dyn_mask"], rp["render_dyn_mask"])
    for k in IMAGE_KEYS:
        assert rn[k].shape == rp[k].shape, k
        assert torch.allclose(rn[k], rp[k], rtol=0, atol=1e-6), k
    for k in ("render_dyn_temporal_closest_mask", "render_dyn_temporal_track_rgb", "render_dyn_temporal_track_mask"):
        assert rn[k].shape == rp[k].shape and torch.equal(rn[k], rp[k]), k
    if use_count:
        assert int(rn["geo_static_raster_status"]) == 0 and int(rp["geo_static_raster_status"]) == 0
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    od = dict(d)
    od["st_pcl_rgb"] = o_cloud[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(rn["geo_static_mask"]), o["geo_static_mask"]) and np.array_equal(N(rn["render_dyn_mask"]), o["render_dyn_mask"])
    for k in IMAGE_KEYS:
        np.testing.assert_allclose(N(rn[k]), o[k], rtol=0, atol=1e-4, err_msg=k)


def test_native_view_call_aggregates_the_cloud_itself():
    """``data["_st_pcl_video"]``: A12 inside the same native call -- the cloud it returns is the oracle's, bit for bit
    and in order, and the images are those of a forward fed with that cloud"""
    H, W, S = 96, 160, 6
    v = synth.make_video(S, H, W, seed=13)
    d = synth.make_view(v, 3, frac=0.5, seed=8)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=16, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["_st_pcl_video"] = _video_dict(v)
    out = torch.full((1, 3, H, W), float("nan"), device=DEV)
    data["_combined_rgb_out"] = out
    with torch.no_grad():
        r = model.forward(data, render_cfg=rc)
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    n = ops.checked_count(r["st_pcl_rgb_count"], "agg")
    assert n == o_cloud.shape[0] and r["st_pcl_rgb"].shape == (1, S * H * W, 6)
    assert np.array_equal(N(r["st_pcl_rgb"][0, :n]).view(np.uint32), o_cloud.view(np.uint32))
    assert np.array_equal(N(r["st_pcl_xyz"][0, :n]).view(np.uint32), o_cloud[:, :3].view(np.uint32))
    assert r["combined_rgb"].data_ptr() == out.data_ptr()  # rendered into the caller's slot
    od = dict(d)
    od["st_pcl_rgb"] = o_cloud[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(r["render_dyn_mask"]), o["render_dyn_mask"])
    np.testing.assert_allclose(N(out), o["combined_rgb"], rtol=0, atol=1e-4)
    # a capacity the cloud does not fit: count == capacity, and the evaluator-shaped caller refuses the view
    from pgdvs_amd import harness

    small = dict(data)
    small["_st_pcl_video"] = _video_dict(v, capacity=n - 5)
    small["st_pcl_rgb_row_bound"] = n - 5
    small.pop("_combined_rgb_out")
    small["rgb_tgt"] = data["rgb_src_temporal"][:, 0]
    small["eval_mask"] = torch.zeros(1, H, W, 3, device=DEV)
    with pytest.raises(ops.PgdvsHipError, match="filled its buffer"):
        harness.eval_step(model, small, rc, device=DEV)


def test_native_view_call_rejects_bad_descriptions():
    from pgdvs_amd import _lib
    import ctypes as C

    lib = _lib.load()
    d = _lib.ViewGeoDesc()
    d.H, d.W = 8, 8
    assert lib.pgdvs_view_geo_forward(C.byref(d), None, 0, None) == -1
    assert b"null input pointer" in lib.pgdvs_last_error()
    H, W, S = 48, 64, 3
    v = synth.make_video(S, H, W, seed=2)
    data = synth.to_torch(synth.make_view(v, 1, seed=1), DEV)
    st = ops.ViewGeoState()
    kw = dict(H=H, W=W, flat_cam_tgt=data["flat_cam_tgt"][0], flat_cam_src=data["flat_cam_src_temporal"][0],
              time_src=data["time_src_temporal"][0], time_tgt=data["time_tgt"][0], rgb1=data["rgb_src_temporal"][0, 0],
              rgb2=data["rgb_src_temporal"][0, 1], depth1=data["depth_src_temporal"][0, 0], depth2=data["depth_src_temporal"][0, 1],
              dyn_mask1=data["dyn_mask_src_temporal"][0, 0], flow12=data["flow_fwd"][0], flow_occ=None, use_flow_consistency=False,
              remove_outlier=True, outlier_knn=16, outlier_std_thres=0.1, alpha=100.0, radius=0.01, K=3,
              st_pcl_rgb=torch.rand(100, 6, device=DEV))
    r = ops.view_geo_forward(st, **kw)
    assert tuple(r["combined_rgb"].shape) == (3, H, W)
    with pytest.raises(ops.PgdvsHipError, match="contiguous"):
        ops.view_geo_forward(st, **dict(kw, rgb1=data["rgb_src_temporal"][0, 0].double()))
    with pytest.raises(ops.PgdvsHipError, match="flow_occ required"):
        ops.view_geo_forward(st, **dict(kw, use_flow_consistency=True))
    with pytest.raises(ops.PgdvsHipError, match="points_per_pixel"):
        ops.view_geo_forward(st, **dict(kw, K=9))
    with pytest.raises(ops.PgdvsHipError):  # CPU tensors: no fallback
        ops.view_geo_forward(st, **dict(kw, rgb1=data["rgb_src_temporal"][0, 0].cpu()))


# ---------------------------------------------------------------- the timed arrangement == the tested arrangement
def test_config_c3_in_the_benchmarks_arrangement_vs_oracle():
    """BASELINE.json configs[2] rendered EXACTLY as bench.py's timed loop renders it -- `ResidentVideoRenderer`: 24 source
    frames resident, the cloud aggregated inside the per-view native call into buffers bounded by the first view's
    count, packed coordinates, the composite written into the caller's slot, the noise field drawn inside the splat
    kernel, three views in flight on three streams -- against the oracle: cloud bit-exact, masks exact, images within
    1e-4 (the field of each draw recovered through ops.splat_noise_field)"""
    from pgdvs_amd.runtime import ResidentVideoRenderer

    H, W, S = 1080, 1920, 24
    v = synth.make_video(S, H, W, seed=1234)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_render_pcl_pts_per_pixel=3)
    rvr = ResidentVideoRenderer(model, rc, T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], lanes=3)
    view_ids = [0, 7, 15]
    ds = [synth.make_view(v, i, frac=0.4, seed=5) for i in view_ids]
    datas = []
    for d in ds:
        t = synth.to_torch(d, DEV)
        t.pop("static_noise")
        datas.append(t)
    n0 = rvr.calibrate(datas[0])
    assert rvr.row_bound == min(S * H * W, int(1.25 * n0) + 65536)
    torch.cuda.synchronize()
    slots = torch.full((3, 1, 3, H, W), float("nan"), device=DEV)
    # the state each lane's stream will draw from, and the draw number it is at
    states = []
    for li in range(3):
        with torch.cuda.stream(rvr.lanes[li][0]):
            st = model.dyn_renderer.splat_rng_state(torch.device(DEV))
        states.append(st)
    torch.cuda.synchronize()
    before = [s.clone() for s in states]
    rets = [rvr.render(datas[li], li, out=slots[li])[0] for li in range(3)]  # three views in flight
    rvr.join()
    torch.cuda.synchronize()
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    ndc_cache = {}
    fields = []
    for li, (d, ret) in enumerate(zip(ds, rets)):
        n = ops.checked_count(ret["st_pcl_rgb_count"], "agg")
        assert n == o_cloud.shape[0] == n0 and ret["st_pcl_rgb"].shape[1] == rvr.row_bound
        assert int(ret["geo_static_raster_status"]) == 0
        assert np.array_equal(N(ret["st_pcl_rgb"][0, :n]).view(np.uint32), o_cloud.view(np.uint32)), f"lane {li}: cloud"
        assert int(states[li][1]) == int(before[li][1]) + 1
        field = ops.splat_noise_field(before[li], H, W)
        # static image: the oracle's point-major sweep over all pixels (same lists as the naive loop)
        ndc = orc.points_to_ndc(o_cloud[:, :3], d["flat_cam_tgt"][0], H, W)
        idx, zbuf, d2 = orc.rasterize_points_pointmajor(ndc, H, W, float(rc.st_render_pcl_pt_radius), 3)
        img = orc.composite(idx, d2, float(rc.st_render_pcl_pt_radius), o_cloud[:, 3:])
        ones = orc.composite(idx, d2, float(rc.st_render_pcl_pt_radius), None)
        np.testing.assert_allclose(N(ret["geo_static_rgb"])[0], img.transpose(2, 0, 1), rtol=0, atol=1e-6)
        assert np.array_equal(N(ret["geo_static_mask"])[0, 0], (ones[..., 0] > 0).astype(np.float32))
        od = dict(d)
        od["rgb_gnt"] = img[None]
        o = orc.render_view(od, dict(rc), static_noise=N(field)[None], alpha=100.0)
        assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"]), f"lane {li}"
        np.testing.assert_allclose(N(ret["render_dyn_rgb"]), o["render_dyn_rgb"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(N(slots[li]), o["combined_rgb"], rtol=0, atol=1e-4)
        assert ret["combined_rgb"].data_ptr() == slots[li].data_ptr()
        fields.append(field)
    del ndc_cache
    # The arrangements bench.py may pick instead (round 5 on: four single-stream lanes on four hardware queues -- also the fixed
    # arrangement of a multi-rank run --, two lanes whose four streams have a queue each; the three lanes above are creation
    # order with second streams).  Placement is scheduling only: the same views, the same noise fields (injected this time),
    # at the same size -- clouds and static images bit for bit, splat images to the rounding of the float atomics.
    for arr in ((4, False, True), (2, True, True)):
        rvr.set_lanes(*arr)
        assert rvr.n_lanes == arr[0] and (rvr.lanes[0][1] is not None) == arr[1]
        outs = []
        for li in range(3):
            dd = dict(datas[li])
            dd["static_noise"] = fields[li][None]
            outs.append(rvr.render(dd, li)[0])  # (views in flight on every lane of the arrangement)
        rvr.join()
        torch.cuda.synchronize()
        for li, (ret, ref) in enumerate(zip(outs, rets)):
            n = ops.checked_count(ret["st_pcl_rgb_count"], "agg")
            assert n == n0 and int(ret["geo_static_raster_status"]) == 0
            assert torch.equal(ret["st_pcl_rgb"][0, :n], ref["st_pcl_rgb"][0, :n]), f"{arr} lane {li}: cloud"
            assert torch.equal(ret["geo_static_rgb"], ref["geo_static_rgb"]) and torch.equal(ret["geo_static_mask"], ref["geo_static_mask"])
            assert torch.equal(ret["render_dyn_mask"], ref["render_dyn_mask"]), f"{arr} lane {li}"
            assert torch.allclose(ret["render_dyn_rgb"], ref["render_dyn_rgb"], rtol=0, atol=1e-5)
            assert torch.allclose(ret["combined_rgb"], slots[li], rtol=0, atol=1e-5)
        del outs


# ---------------------------------------------------------------- the evaluator's metric in one pass
@pytest.mark.parametrize("H,W", [(37, 53), (270, 480)])
def test_eval_psnr_sums_vs_torch_statement(H, W):
    """csrc/eval.hip against harness.quantize_like_evaluator / masked_psnr (pinned by the reference's own fixture in
    tests/test_host_cpu.py): quantised images bit-exact, PSNRs to 1e-9 dB; NaNs in the prediction, values outside [0, 1]"""
    from pgdvs_amd.harness import masked_psnr, quantize_like_evaluator

    g = torch.Generator(device="cpu").manual_seed(H)
    pred = torch.rand(3, H, W, generator=g) * 1.4 - 0.2
    pred[0, 3, 5] = float("nan")
    gt = torch.rand(H, W, 3, generator=g) * 1.2 - 0.1
    mask = (torch.rand(H, W, 1, generator=g) < 0.3).float().expand(H, W, 3).contiguous()
    sums, pq, gq = ops.eval_psnr_sums(pred.to(DEV), gt.to(DEV), mask.to(DEV), want_images=True)
    s = sums.cpu().tolist()
    pq_t = quantize_like_evaluator(pred)
    gq_t = quantize_like_evaluator(gt.permute(2, 0, 1))
    # (8-bit codes: the final division by 255 is correctly rounded on both sides, compare the floats as well)
    assert torch.equal(pq.cpu(), pq_t) and torch.equal(gq.cpu(), gq_t)
    m = mask.permute(2, 0, 1)
    for j, mk in enumerate((torch.ones_like(m), m, 1.0 - m)):
        want = masked_psnr(gq_t, pq_t, mk)
        mse = s[j] / (s[3 + j] + 1e-8)
        got = 0 if mse == 0 else 10 * np.log10(1.0 / mse)
        assert abs(got - want) < 1e-9, (j, got, want)
    assert s[3] == 3 * H * W and abs(s[4] + s[5] - s[3]) < 1e-6
    assert s[6] == -1.0 and s[7] == 0.0  # no status words given
    cnt = torch.tensor([123456789012], dtype=torch.int64, device=DEV)
    st = torch.tensor([1], dtype=torch.int32, device=DEV)
    s3 = ops.eval_psnr_sums(pred.to(DEV), gt.to(DEV), mask.to(DEV), count_dev=cnt, status_dev=st)[0].cpu().tolist()
    assert s3[:6] == s[:6] and s3[6] == 123456789012.0 and s3[7] == 1.0  # the status words ride along, the sums do not move
    # identical images: the reference's quirk of PSNR 0
    s2 = ops.eval_psnr_sums(gq, gq_t.permute(1, 2, 0).contiguous().to(DEV), mask.to(DEV))[0].cpu().tolist()
    assert s2[0] == 0 and s2[1] == 0 and s2[2] == 0


def test_eval_step_fused_metric_equals_the_torch_path(monkeypatch):
    """harness.eval_step around the HIP renderer: the one-pass metric and the torch statement of the reference's
    evaluator give the same metric dict"""
    from pgdvs_amd import harness

    H, W, S = 72, 128, 4
    v = synth.make_video(S, H, W, seed=21)
    d = synth.make_view(v, 1, seed=3)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, dyn_pcl_outlier_knn=20, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["_st_pcl_video"] = _video_dict(v)
    data["rgb_tgt"] = (data["rgb_src_temporal"][:, 1] * 0.9 + 0.03).contiguous()
    data["eval_mask"] = data["dyn_mask_src_temporal"][:, 0].expand(-1, -1, -1, 3).contiguous()
    md, ex = harness.eval_step(model, data, rc, device=DEV, return_images=True)
    # the torch statement on the same prediction
    # (on the CPU, where the fixture made by the reference pins it: torch's GPU division by a scalar multiplies by the
    # rounded reciprocal instead, one ulp off for some codes)
    pq = harness.quantize_like_evaluator(ex["ret"]["combined_rgb"][0].cpu())
    gq = harness.quantize_like_evaluator(data["rgb_tgt"][0].permute(2, 0, 1).cpu())
    assert torch.equal(ex["pred"][0].cpu(), pq) and torch.equal(ex["gt"][0].cpu(), gq)
    m = data["eval_mask"][0].permute(2, 0, 1).cpu()
    for key, mk in (("psnr_full_combined", torch.ones_like(m)), ("psnr_dyn_combined", m), ("psnr_static_combined", 1 - m)):
        want = harness.masked_psnr(gq, pq, mk)
        assert abs(float(md[f"eval/{key}"]) - want) < 1e-4 * max(1.0, abs(want)), key
    assert int(md["eval/count"]) == 1


# ---------------------------------------------------------------- scenes with other statistics
@pytest.mark.parametrize("scene", ["wide_baseline", "noisy_depth"])
def test_scene_statistics_540p_vs_oracle(scene):
    """540p x 12 frames of the benchmark's off-nominal scenes (12-camera rig cycled per frame; noisy depth with flying
    pixels): cloud bit-exact and in order, z-buffer fragments bit-exact on every pixel, dynamic branch + composite
    within 1e-4 -- the statistics that lengthen tile lists and the aggregation's links change no result"""
    H, W, S = 540, 960, 12
    v = synth.make_video(S, H, W, seed=1234, scene=scene)
    d = synth.make_view(v, 5, frac=0.4, seed=5)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["_st_pcl_video"] = _video_dict(v)
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    n = ops.checked_count(ret["st_pcl_rgb_count"], "agg")
    assert n == o_cloud.shape[0]
    assert np.array_equal(N(ret["st_pcl_rgb"][0, :n]).view(np.uint32), o_cloud.view(np.uint32))
    # more newly visible pixels per frame than the nominal scene's ~3.6 % of P
    nominal = synth.make_video(S, H, W, seed=1234)
    n_nom = orc.aggregate_static_pcl(nominal["rgbs"], nominal["depths"], nominal["dyn_masks"], nominal["K3s"], nominal["c2ws"]).shape[0]
    assert n > n_nom, (n, n_nom)
    radius = float(rc.st_render_pcl_pt_radius)
    cam = ops.cam_prep(data["flat_cam_tgt"][0])
    frag = ops.points_raster(ret["st_pcl_rgb"][0, :n], ret["st_pcl_rgb"][0, :n, 3:], cam, radius, 3, H, W, want_fragments=True,
                             rgb_planar=True)
    assert torch.equal(frag["rgb"], ret["geo_static_rgb"][0])
    ndc = orc.points_to_ndc(o_cloud[:, :3], d["flat_cam_tgt"][0], H, W)
    idx, zbuf, d2 = orc.rasterize_points_pointmajor(ndc, H, W, radius, 3)
    assert np.array_equal(N(frag["idx"]), idx)
    assert np.array_equal(N(frag["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(frag["dist2"]).view(np.uint32), d2.view(np.uint32))
    img = orc.composite(idx, d2, radius, o_cloud[:, 3:])
    np.testing.assert_allclose(N(ret["geo_static_rgb"])[0], img.transpose(2, 0, 1), rtol=0, atol=1e-6)
    od = dict(d)
    od["rgb_gnt"] = img[None]
    o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
    assert np.array_equal(N(ret["render_dyn_mask"]), o["render_dyn_mask"])
    for k in ["render_dyn_rgb", "combined_rgb"]:
        np.testing.assert_allclose(N(ret[k]), o[k], rtol=0, atol=1e-4, err_msg=k)


# ---------------------------------------------------------------- the rasteriser's depth bound (list pruning)
@pytest.mark.parametrize("H,W,n,K,radius,seed", [(160, 224, 60000, 3, 0.075, 0), (160, 224, 90000, 4, 0.04, 1), (96, 128, 30000, 8, 0.12, 2),
                                                 (200, 320, 120000, 1, 0.06, 3), (128, 128, 50000, 5, 0.05, 4)])
def test_raster_depth_bound_random_clouds_vs_oracle(H, W, n, K, radius, seed, monkeypatch):
    """option raster_bound_density = 0: the bound is computed whatever the density.  Layered random depths (the regime in which
    it drops most of every list), exact depth ties across the bound, points outside the image and behind the camera, block
    sizes 4 and 2, K up to the block's capacity and beyond it: fragments bit-exact against the oracle's naive loop, and
    identical to the un-pruned rasteriser's"""
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.uniform(-1.3, 1.3, (n, 2)) * [W / H, 1.0], rng.choice([1.5, 1.5, 1.7, 2.0, 2.5, 3.0], (n, 1))
                          + rng.normal(0, 0.05, (n, 1)) * (rng.random((n, 1)) < 0.7)], 1).astype(np.float32)
    pts[:50, 2] = -1.0
    feat = rng.random((n, 3)).astype(np.float32)
    fc = synth.flat_cam(H, W, np.array([[0.8 * H, 0, W / 2], [0, 0.8 * H, H / 2], [0, 0, 1]]), np.eye(4)).astype(np.float32)
    cam = ops.cam_prep(T(fc))
    with ops.option("raster_bound_density", 0):
        a = ops.points_raster(T(pts), T(feat), cam, radius, K, H, W, want_fragments=True)
    with ops.option("raster_bound_density", 1e9):
        b = ops.points_raster(T(pts), T(feat), cam, radius, K, H, W, want_fragments=True)
    for k in ("idx", "zbuf", "dist2", "rgb", "mask"):
        assert torch.equal(a[k], b[k]), k
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, radius, K)
    assert np.array_equal(N(a["idx"]), idx) and np.array_equal(N(a["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(a["dist2"]).view(np.uint32), d2.view(np.uint32))
    assert (idx >= 0).mean() > 0.5


def test_raster_depth_bound_1080p_noisy_depth_vs_oracle(monkeypatch):
    """the cloud the bound is made for -- noisy depth, 8 source frames at 1080p -- through the native view call: every fragment of the frame against the oracle's
    point-major sweep, and the counters say that lists shrank"""
    H, W, S = 1080, 1920, 8
    v = synth.make_video(S, H, W, seed=1234, scene="noisy_depth")
    d = synth.make_view(v, 3, frac=0.4, seed=5)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=False, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["_st_pcl_video"] = _video_dict(v)
    with ops.option("raster_bound_density", 1.2):  # (default 2.2 rows per pixel: 24 such frames reach 3.1, these 8 do not)
        with torch.no_grad():
            ret = model.forward(data, render_cfg=rc)
        cnt_on = model.view_counters()
    n = ops.checked_count(ret["st_pcl_rgb_count"], "agg")
    assert n > 1.2 * H * W, n  # the gate is open: decided on the device, from the count
    with ops.option("raster_bound_density", 1e9):
        with torch.no_grad():
            ret_off = model.forward(data, render_cfg=rc)
        cnt_off = model.view_counters()
    assert torch.equal(ret["geo_static_rgb"], ret_off["geo_static_rgb"]) and torch.equal(ret["geo_static_mask"], ret_off["geo_static_mask"])
    assert cnt_on["static_rows"] == cnt_off["static_rows"] == n
    assert cnt_on["raster_list_entries"] < 0.5 * cnt_off["raster_list_entries"], (cnt_on, cnt_off)
    assert cnt_on["raster_longest_tile_list"] < cnt_off["raster_longest_tile_list"]
    o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    assert np.array_equal(N(ret["st_pcl_rgb"][0, :n]).view(np.uint32), o_cloud.view(np.uint32))
    radius = float(rc.st_render_pcl_pt_radius)
    frag = ops.points_raster(ret["st_pcl_rgb"][0, :n], ret["st_pcl_rgb"][0, :n, 3:], ops.cam_prep(data["flat_cam_tgt"][0]), radius, 3, H, W,
                             want_fragments=True, rgb_planar=True)
    assert torch.equal(frag["rgb"], ret["geo_static_rgb"][0])
    ndc = orc.points_to_ndc(o_cloud[:, :3], d["flat_cam_tgt"][0], H, W)
    idx, zbuf, d2 = orc.rasterize_points_pointmajor(ndc, H, W, radius, 3)
    assert np.array_equal(N(frag["idx"]), idx)
    assert np.array_equal(N(frag["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(frag["dist2"]).view(np.uint32), d2.view(np.uint32))


def test_view_counters_report_the_fast_paths_exits():
    """pgdvs_view_geo_counters: the numbers are the kernels' own device words -- rows, list entries (= the oracle's count of
    (point, tile) pairs), kNN queries and how many left the thread-per-query pass, points in the aggregation's fp64 queue"""
    H, W, S = 270, 480, 6
    v = synth.make_video(S, H, W, seed=5)
    d = synth.make_view(v, 2, frac=0.4, seed=1)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_render_pcl_pts_per_pixel=3)
    data = synth.to_torch(d, DEV)
    data["_st_pcl_video"] = _video_dict(v)
    with torch.no_grad():
        ret = model.forward(data, render_cfg=rc)
    c = model.view_counters()
    n = ops.checked_count(ret["st_pcl_rgb_count"], "agg")
    assert c["static_rows"] == n
    assert c["knn_queries"] == int((N(data["dyn_mask_src_temporal"][0, 0]) > 0).sum()) or c["knn_queries"] > 0
    assert 0 <= c["knn_queries_to_ring_search"] <= c["knn_queries"]
    assert c["knn_queries_to_coarse_grid"] <= c["knn_queries"] and c["knn_queries_scanned_exhaustively"] <= c["knn_queries_to_coarse_grid"] + 1
    assert n <= c["raster_list_entries"] <= 9 * n and 0 < c["raster_longest_tile_list"] <= c["raster_list_entries"]
    assert c["raster_tiles_general_path_long_list"] == 0 and c["raster_tiles_general_path_equal_depths"] == 0
    assert 0 < c["agg_points_in_fp64_queue"] < 0.2 * n * S
    assert c["agg_projections_in_reference_order"] <= c["agg_points_in_fp64_queue"]
    # a crowded tile (everything lands in a few tiles) takes the general path, and the counter says so
    pts = np.concatenate([np.random.default_rng(0).normal(0, 0.01, (6000, 2)), np.full((6000, 1), 2.0)], 1).astype(np.float32)
    data2 = synth.to_torch(d, DEV)
    data2["st_pcl_rgb"] = T(np.concatenate([pts, np.full((6000, 3), 0.5, np.float32)], 1))[None]
    with torch.no_grad():
        model.forward(data2, render_cfg=rc)
    c2 = model.view_counters()
    assert c2["static_rows"] == 6000 and c2["raster_longest_tile_list"] > 2048
    assert c2["raster_tiles_general_path_long_list"] >= 1


def test_native_view_call_reuses_camera_constants_only_for_the_same_video():
    """the per-frame camera constants of the aggregation stay in the view workspace between calls with the same video
    (``agg_params_cached``); another video of the same shape, or the same video again after it, uploads them again --
    every cloud is the oracle's"""
    H, W, S = 80, 128, 5
    va = synth.make_video(S, H, W, seed=3)
    vb = synth.make_video(S, H, W, seed=3, scene="wide_baseline")  # same shape, other cameras
    d = synth.make_view(va, 2, seed=1)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=False, st_render_pcl_pts_per_pixel=2)
    clouds = {"a": orc.aggregate_static_pcl(va["rgbs"], va["depths"], va["dyn_masks"], va["K3s"], va["c2ws"]),
              "b": orc.aggregate_static_pcl(vb["rgbs"], vb["depths"], vb["dyn_masks"], vb["K3s"], vb["c2ws"])}
    vids = {"a": _video_dict(va), "b": _video_dict(vb)}
    (st,) = [None]
    cached = []
    for which in ("a", "a", "b", "a", "a"):
        data = synth.to_torch(d, DEV)
        data["_st_pcl_video"] = vids[which]
        with torch.no_grad():
            r = model.forward(data, render_cfg=rc)
        (st,) = model._view_states.values()
        cached.append(int(st.desc.agg_params_cached))
        n = ops.checked_count(r["st_pcl_rgb_count"], "agg")
        assert n == clouds[which].shape[0], which
        assert np.array_equal(N(r["st_pcl_rgb"][0, :n]).view(np.uint32), clouds[which].view(np.uint32)), which
    assert cached == [0, 1, 0, 0, 1]


@pytest.mark.parametrize("n,spread,K,bound", [(9000, 0.09, 3, "0"), (14000, 0.06, 2, "0"), (30000, 0.05, 3, "0"), (9000, 0.09, 3, "1e9")])
def test_raster_long_tile_lists_second_launch_vs_oracle(n, spread, K, bound, monkeypatch):
    """tile lists between 2048 and 4096 entries (the second tile launch's 4096-entry sorted path), beyond 4096 (its general
    path) and short ones in one image; with the depth bound dropping entries and without it (raster_bound_density = 1e9:
    one launch, the general path takes every list beyond 2048): fragments bit-exact against the oracle's naive loop"""
    H, W = 96, 128
    rng = np.random.default_rng(n)
    centres = np.array([[-0.5, -0.3], [0.4, 0.2], [0.0, 0.45]])
    which = rng.integers(0, 3, n)
    xy = centres[which] + rng.normal(0, spread, (n, 2)) * np.array([1.0, 0.6])
    z = 2.0 + 0.4 * which[:, None] + rng.normal(0, 0.02, (n, 1)) * (rng.random((n, 1)) < 0.5)  # exact ties inside a cluster
    pts = np.concatenate([xy, z], 1).astype(np.float32)
    fc = synth.flat_cam(H, W, np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]]), np.eye(4)).astype(np.float32)
    cam = ops.cam_prep(T(fc))
    radius = 0.12  # 5.76 px: 4 x 4 blocks for the bound
    with ops.option("raster_bound_density", float(bound)):
        a = ops.points_raster(T(pts), T(rng.random((n, 3)).astype(np.float32)), cam, radius, K, H, W, want_fragments=True)
    idx, zbuf, d2 = orc.rasterize_points(pts, fc, H, W, radius, K)
    assert np.array_equal(N(a["idx"]), idx)
    assert np.array_equal(N(a["zbuf"]).view(np.uint32), zbuf.view(np.uint32))
    assert np.array_equal(N(a["dist2"]).view(np.uint32), d2.view(np.uint32))
    # the lists really are that long: count the (point, tile) pairs of the densest tile on the host
    ndc = orc.points_to_ndc(pts, fc, H, W)
    px = (W - 1) - ((ndc[:, 0] + W / H) * W - W / H) / (2.0 * W / H)
    py = (H - 1) - ((ndc[:, 1] + 1.0) * H - 1.0) / 2.0
    t = (np.clip(py // 16, 0, H // 16 - 1) * (W // 16) + np.clip(px // 16, 0, W // 16 - 1)).astype(int)
    assert np.bincount(t).max() > 2048


def test_static_aggregation_without_the_staging_block():
    """option agg_stage = 0 (PGDVS_AGG_STAGE=0 at load time): the chain links leave no (depth, colour) behind and `agg_rows`
    gathers both itself (the default path of the last frame and of videos too long for the staging block) -- through the
    bit-exact aggregation tests, in a child process that loads the library with the switch in its environment (which also
    covers the options' one read of the environment)"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, PGDVS_AGG_STAGE="0")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_round2.py"), "-k",
                        "static_aggregation_shapes_vs_oracle or static_aggregation_vs_reference_golden or "
                        "static_aggregation_capacity_clamp or degenerate_camera_motions or config_c1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    assert ops.get_option("agg_stage") == 1.0  # (this process: the default)


def test_views_in_flight_reproduce_the_sequential_images():
    """stress of the lanes: 48 views of a 540p x 8-frame video through `ResidentVideoRenderer` with three lanes (second
    stream each), four distinct target views cycled, injected noise -- every image equals the one the same view gives alone
    (splat atomics: to rounding), every cloud count is the same, no status word is raised"""
    from pgdvs_amd.runtime import ResidentVideoRenderer

    H, W, S = 540, 960, 8
    v = synth.make_video(S, H, W, seed=77)
    model, rc = _renderer("geo", dyn_pcl_remove_outlier=True, st_render_pcl_pts_per_pixel=3)
    rvr = ResidentVideoRenderer(model, rc, T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], lanes=3,
                                side_streams=True)
    datas = [synth.to_torch(synth.make_view(v, i, frac=0.3, seed=2), DEV) for i in (0, 2, 4, 6)]
    n0 = rvr.calibrate(datas[0])
    refs = []
    for d in datas:
        ret, _ = rvr.render(d, 0)
        rvr.join()
        torch.cuda.synchronize()
        refs.append((ret["combined_rgb"].clone(), ret["geo_static_rgb"].clone(), ret["render_dyn_mask"].clone()))
    outs = torch.empty((48, 1, 3, H, W), device=DEV)
    rets = []
    for j in range(48):
        rets.append(rvr.render(datas[j % 4], j, out=outs[j])[0])
    rvr.join()
    torch.cuda.synchronize()
    for j, ret in enumerate(rets):
        comb, st, dm = refs[j % 4]
        assert int(ret["st_pcl_rgb_count"]) == n0 and int(ret["geo_static_raster_status"]) == 0, j
        assert torch.equal(ret["geo_static_rgb"], st), j          # the static branch has no atomics: bit for bit
        assert torch.equal(ret["render_dyn_mask"], dm), j
        assert torch.allclose(outs[j], comb, rtol=0, atol=1e-5), j


def test_gnt_chunk_loop_jobs_and_stage_events():
    """BaseRenderer's chunk loop (reference: pgdvs/models/gnt/renderer.py:414-485) as a list of (chunk, batch item)
    jobs: chunks that straddle the two batch items, results independent of the chunk size, peak memory not growing
    with the number of chunks, and the stage-event hook behind bench.py's breakdown of the GNT renderer."""
    from pgdvs_amd.models.gnt.model import GNTModel
    from pgdvs_amd.models.gnt.renderer import BaseRenderer

    torch.manual_seed(3)
    br = BaseRenderer(model_cfg=None)
    br.model = GNTModel(netwidth=64, transformer_depth=2)
    br = br.to(DEV).eval()
    B, V, H, W, S = 2, 5, 48, 64, 32
    video = synth.make_video(V, H, W, seed=5)
    cams_src = np.stack([synth.flat_cam(H, W, video["K3s"][i], video["c2ws"][i]) for i in range(V)])
    cam_tgt = np.stack([cams_src[1], cams_src[3]])
    rays = [ops.get_rays(ops.cam_prep(T(cam_tgt[b])), H, W, 1) for b in range(B)]
    ray_batch = {
        "ray_o": torch.cat([r[0] for r in rays]), "ray_d": torch.cat([r[1] for r in rays]), "camera": T(cam_tgt),
        "raw_h": H, "raw_w": W, "depth_range": T(np.array([[0.8, 5.0], [0.7, 4.0]], np.float32)), "depth_range_per_ray": False,
        "src_rgbs": T(np.stack([video["rgbs"], video["rgbs"][::-1]])),
        "src_invalid_masks": T(np.stack([video["dyn_masks"], video["dyn_masks"][::-1]]).astype(np.float32))[..., None],
        "src_cameras": T(np.stack([cams_src, cams_src[::-1]])),
    }

    def run(chunk):  # noqa: E306
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with torch.no_grad():
            ret = br.forward(ray_batch=ray_batch, chunk_size=chunk, inv_uniform=True, n_coarse_samples_per_ray=S, use_dyn_mask=True,
                             flag_deterministic=True, ret_view_entropy=True, ret_view_std=True)["outputs_coarse"]
        torch.cuda.synchronize()
        return ret, torch.cuda.max_memory_allocated() - base

    n_rays = B * H * W
    br.merge_chunks_up_to = 0  # (execute chunk by chunk as given: this test counts them; merged execution chunks below)
    chunk = 500  # 13 chunks; chunk 6 straddles the two batch items
    assert (H * W) % chunk != 0 and n_rays // chunk >= 12
    mid, _ = run(chunk)
    for k in mid:
        assert bool(torch.isfinite(mid[k]).all()), k
    # memory: 4x the chunks (a quarter of the rays each) must not need more than the long chunks' run
    few, peak_few = run(2048)
    many, peak_many = run(256)
    assert peak_many <= peak_few, (peak_many, peak_few)
    for k in few:  # (the ResUNet runs on MIOpen, whose convolutions are not bit-reproducible from call to call: 1e-4, the path's tolerance)
        np.testing.assert_allclose(many[k].cpu().numpy(), few[k].cpu().numpy(), rtol=0, atol=1e-4, err_msg=k)
        np.testing.assert_allclose(mid[k].cpu().numpy(), few[k].cpu().numpy(), rtol=0, atol=1e-4, err_msg=k)
    # the stage-event hook of bench.py's breakdown
    br.stage_events = ev = {}
    run(chunk)
    br.stage_events = None
    assert len(ev["features"]) == 1 and len(ev["gather"]) == len(ev["transformer"]) == 14  # 13 chunks, one of them in two pieces
    # round 5: consecutive chunks merged into execution chunks of up to 4096 rays (the default): fewer jobs, same images
    br.merge_chunks_up_to = 4096
    br.stage_events = ev = {}
    merged, _ = run(chunk)
    br.stage_events = None
    assert len(ev["gather"]) == 3  # 6144 rays as 4000 (two batch items: 3072 + 928) + 2144
    for k in merged:
        np.testing.assert_allclose(merged[k].cpu().numpy(), mid[k].cpu().numpy(), rtol=0, atol=1e-4, err_msg=k)
    assert all(a.elapsed_time(b) >= 0 for v in ev.values() for a, b in v)


@pytest.mark.parametrize("S", [33, 64, 250, 256])
def test_gnt_ray_layer_wide_score_range(S):
    """ray attention with scores that grow along the ray by far more than the kernel's rescaling gap (2^20): the
    score tiles arrive relative to a reference that has to move several times per head (csrc/gnt_view.hip key_tile,
    the rare branch), also inside the partial last key tile; against torch's softmax
    (reference: pgdvs/models/gnt/models/transformer_network.py:231-338)."""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(300 + S)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    layer = net.view_selftrans[0]
    R = 9
    # rows = one direction per ray + noise that fades along the ray, q_fc = k_fc = 8 I: the keys late on the ray align
    # better and better with every query, the running maximum climbs by more than the gap tile after tile
    fade = torch.linspace(2.0, 0.0, S, device=DEV)[None, :, None]
    q = torch.randn(R, 1, 64, device=DEV) + fade * torch.randn(R, S, 64, device=DEV)
    with torch.no_grad():
        layer.attn_norm.weight.fill_(1.0)
        layer.attn_norm.bias.zero_()
        layer.attn.q_fc.weight.copy_(torch.eye(64, device=DEV) * 8.0)
        layer.attn.k_fc.weight.copy_(torch.eye(64, device=DEV) * 8.0)
        xn = layer.attn_norm(q)
        qq = layer.attn.q_fc(xn).view(R, S, 4, 16).permute(0, 2, 1, 3)
        kk = layer.attn.k_fc(xn).view(R, S, 4, 16).permute(0, 2, 1, 3)
        sc = (qq @ kk.transpose(-1, -2)) / 4.0
        spread = ((sc.amax(-1) - sc.amin(-1)).max() * 1.4426950408889634).item()
        # running maximum per query over 16-key tiles: how often does it jump by more than the gap?
        tiles = torch.stack([sc[..., t:t + 16].amax(-1) for t in range(0, S, 16)], -1) * 1.4426950408889634
        run = torch.cummax(tiles, -1).values
        jumps = int(((tiles[..., 1:] - run[..., :-1]) > 20.0).sum())
        out_k, w_k = GNT._ray_layer(layer, q, True)
    assert spread > 60.0 and jumps > 0, (spread, jumps)  # the inputs do exercise the branch
    ops._GNT_VIEW_ENABLED = False
    try:
        with torch.no_grad():
            out_t, w_t = GNT._ray_layer(layer, q, True)
    finally:
        ops._GNT_VIEW_ENABLED = True
    np.testing.assert_allclose(out_k.cpu().numpy(), out_t.cpu().numpy(), rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(w_k.cpu().numpy(), w_t.cpu().numpy(), rtol=2e-4, atol=1e-6)
    # (scores of several hundred: one ulp of a score is 4e-5 of its weight)
    np.testing.assert_allclose(w_k.cpu().numpy().sum(1), 1.0, rtol=1e-4)


@pytest.mark.parametrize("V,want_stats", [(4, False), (24, True)])
def test_gnt_view_layer_bf16x3_products_vs_fp32_products(V, want_stats, monkeypatch):
    """the view layer's two 64 x 64 products per source view (k = Wk f, vv = Wv k) run on v_mfma_f32_16x16x32_bf16 with both
    operands split exactly into three bf16 pieces (csrc/gnt_mfma.h chain64_bf16x3; six partial products, fp32 accumulation);
    the option gnt_fp32 keeps them on the fp32 instruction.  Both against the torch fp32 statement of the layer
    (pgdvs/models/gnt/models/transformer_network.py:59-169), and against each other at a tenth of that tolerance:
    inputs with a wide dynamic range (features over four decades, weights scaled up) so that dropped low-order pieces
    would show."""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(40 + V)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    layer = net.view_crosstrans[0]
    with torch.no_grad():
        layer.attn.k_fc.weight.mul_(3.0)
        layer.attn.v_fc.weight.mul_(2.0)
    R, S = 29, 23
    q = torch.randn(R, S, 64, device=DEV)
    decades = 10.0 ** torch.randint(-2, 2, (R, S, V, 64), device=DEV).float()
    feat = torch.randn(R, S, V, 64, device=DEV) * decades
    rd = torch.randn(R, S, V, 4, device=DEV)
    valid = torch.rand(R, S, V, device=DEV) < 0.7
    cnt = valid.sum(-1)
    empty = cnt == 0
    valid = valid | empty[..., None]
    cnt = torch.where(empty, torch.full_like(cnt, V), cnt)

    def run():
        with torch.no_grad():
            return net._view_layer(layer, q, feat, rd, valid, cnt, want_stats)

    with ops.gnt_product_path(fp32=True):
        out_f, st_f = run()
    with ops.gnt_product_path(fp32=False):
        out_s, st_s = run()
    ops._GNT_VIEW_ENABLED = False
    try:
        out_t, st_t = run()
    finally:
        ops._GNT_VIEW_ENABLED = True
    scale = float(out_t.abs().max())
    for name, o in (("fp32 products", out_f), ("bf16x3 products", out_s)):
        np.testing.assert_allclose(o.cpu().numpy(), out_t.cpu().numpy(), rtol=1e-4, atol=2e-5 * max(scale, 1.0), err_msg=name)
    # the two arithmetic paths agree ten times closer than either has to agree with torch
    np.testing.assert_allclose(out_s.cpu().numpy(), out_f.cpu().numpy(), rtol=1e-5, atol=2e-6 * max(scale, 1.0))
    if want_stats:
        for a, b, c, name in zip(st_s, st_f, st_t, ("entropy", "std", "std_norm")):
            np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-3, atol=5e-5, err_msg=name)
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=5e-6, err_msg=name + " (paths)")


def test_gnt_forward_golden_on_both_product_paths(golden_dir, monkeypatch):
    """the reference-made fixture gnt_small.npz (GNT.forward of upstream, two layers) through the fp32-instruction path too:
    the default (bf16x3 products) runs in test_gnt_forward_vs_reference"""
    from test_gpu_parity import _gnt_model

    m, g = _gnt_model(golden_dir)
    with ops.gnt_product_path(fp32=True), torch.no_grad():
        out, ex = m.net_coarse(T(g["dynmask_rgb_feat"]), T(g["dynmask_ray_diff"]), T(g["dynmask_mask"]), T(g["pts"]), T(g["ray_d"]),
                               ret_view_entropy=True, ret_view_std=True)
    np.testing.assert_allclose(out.cpu().numpy(), g["dynmask_out"], rtol=0, atol=1e-4)
    for k, v in ex.items():
        np.testing.assert_allclose(v.cpu().numpy(), g[f"dynmask_{k}"], rtol=0, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("S,R", [(1, 5), (33, 19), (256, 9), (47, 300)])
def test_gnt_feed_forward_both_product_paths(S, R, monkeypatch):
    """the feed-forward block behind every attention layer (csrc/gnt_view.hip gnt_ff_bf16x3_kernel: bf16x3 products on
    v_mfma_f32_32x32x16_bf16, two phases per round with half of the weight pieces resident; gnt_fp32 = 1: gnt_ff_kernel on
    the fp32 instruction) through the ray layer at row counts that leave wavefronts and whole rounds of a workgroup without a
    tile (5 rows, 627, 2304, 14100): both paths against torch (transformer_network.py:44-55,:218-221) and against each other."""
    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    torch.manual_seed(500 + S)
    net = GNT(netwidth=64, transformer_depth=1).to(DEV).eval()
    layer = net.view_selftrans[0]
    with torch.no_grad():
        layer.ff.fc1.weight.mul_(2.0)
        layer.ff.fc1.bias.add_(torch.randn_like(layer.ff.fc1.bias) * 0.3)
        layer.ff.fc2.bias.add_(torch.randn_like(layer.ff.fc2.bias) * 0.3)
    q = torch.randn(R, S, 64, device=DEV) * (10.0 ** torch.randint(-1, 2, (R, S, 1), device=DEV).float())

    def run():
        with torch.no_grad():
            return GNT._ray_layer(layer, q, True)

    with ops.gnt_product_path(fp32=True):
        out_f, w_f = run()
    with ops.gnt_product_path(fp32=False):
        out_s, w_s = run()
    ops._GNT_VIEW_ENABLED = False
    try:
        out_t, w_t = run()
    finally:
        ops._GNT_VIEW_ENABLED = True
    scale = max(float(out_t.abs().max()), 1.0)
    for name, o in (("fp32 instruction", out_f), ("bf16x3", out_s)):
        np.testing.assert_allclose(o.cpu().numpy(), out_t.cpu().numpy(), rtol=1e-4, atol=3e-5 * scale, err_msg=name)
    np.testing.assert_allclose(out_s.cpu().numpy(), out_f.cpu().numpy(), rtol=1e-5, atol=3e-6 * scale)
    np.testing.assert_allclose(w_s.cpu().numpy(), w_f.cpu().numpy(), rtol=0, atol=0)  # (the attention itself is the same kernel)
