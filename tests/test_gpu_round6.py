"""GPU tests added in round 6 (MI355X): the per-view call's FUSED launches (csrc/fused.h -- clears, counts, the
compaction with its gather and bounding box, the median's bin selections inside the passes, keep[idx] written by the filter's
last launch, the projection inside the splat's flag pass, (pixel, frame) pairs dealt over all wavefronts of an aggregation link,
tickets in the kNN grid's look-back scan) against the per-op entry points, which keep their own launches, and against the
oracle; the renderer's default second stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)
from pgdvs_amd import ops, synth  # noqa: E402

DEV = "cuda:0"
IMAGE_KEYS = ["geo_static_rgb", "render_dyn_rgb", "combined_rgb", "combined_rgb_static", "combined_rgb_dyn"]


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


def _renderer(**over):
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

    cfg = load_config(static_renderer="geo")
    rc = cfg.engine.engine_cfg.render_cfg
    for k, v in over.items():
        rc[k] = v
    return PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(DEV).eval(), rc


@pytest.mark.parametrize("H,W,S,outlier,knn,scene", [(333, 517, 4, True, 50, "nominal"), (333, 517, 4, False, 50, "nominal"),
                                                     (270, 480, 6, True, 50, "noisy_depth"), (61, 67, 3, True, 8, "nominal"),
                                                     (540, 960, 5, True, 50, "wide_baseline")])
def test_fused_view_call_equals_per_op_path_and_oracle(H, W, S, outlier, knn, scene, monkeypatch):
    """sizes that are no multiple of a chunk (256), of a compaction tile (4096) or of anything else; several compaction
    tiles, so that the offsets come from other workgroups' chunk counts; a cloud from noisy depth (the kNN grid's ring
    search and coarse grid run: grid2_setup, the merged fallback launch); with and without the filter"""
    v = synth.make_video(S, H, W, seed=77, scene=scene)
    d = synth.make_view(v, 1, frac=0.35, seed=9)
    model, rc = _renderer(dyn_pcl_remove_outlier=outlier, dyn_pcl_outlier_knn=knn)
    cloud, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=S * H * W)
    n = ops.checked_count(cnt, "agg")
    data = synth.to_torch(d, DEV)
    data["st_pcl_rgb"] = cloud[None, :n].contiguous()
    with torch.no_grad():
        rn = model.forward(dict(data), render_cfg=rc)  # (the renderer's default second stream)
        rn1 = model.forward(dict(data, _side_stream=False), render_cfg=rc)
        monkeypatch.setenv("PGDVS_NATIVE_VIEW", "0")
        assert not model._native_view_ok(data, rc)
        rp = model.forward(dict(data), render_cfg=rc)
    torch.cuda.synchronize()
    for r in (rn, rn1):
        assert torch.equal(r["geo_static_rgb"], rp["geo_static_rgb"]) and torch.equal(r["geo_static_mask"], rp["geo_static_mask"])
        assert torch.equal(r["render_dyn_mask"], rp["render_dyn_mask"])
        for k in IMAGE_KEYS:
            assert torch.allclose(r[k], rp[k], rtol=0, atol=1e-6), k
    if H * W <= 200 * 1000:  # (the oracle's brute-force kNN and naive rasteriser: seconds at these sizes)
        o_cloud = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
        od = dict(d)
        od["st_pcl_rgb"] = o_cloud[None]
        o = orc.render_view(od, dict(rc), static_noise=d["static_noise"], alpha=100.0)
        assert np.array_equal(N(rn["render_dyn_mask"]), o["render_dyn_mask"])
        for k in IMAGE_KEYS:
            np.testing.assert_allclose(N(rn[k]), o[k], rtol=0, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("S,n_sel", [(5, "few"), (12, "many"), (33, "few")])
def test_aggregation_links_with_parts(S, n_sel):
    """the links' (pixel, frame) pairs dealt over all four wavefronts (1, 2 or 4 parts by the workgroup's list length, more
    than 32 later frames -> several rows of workgroups): clouds bit-exact and in order against the oracle, with few newly
    visible pixels per link (smooth camera path: lists of a few dozen pixels -> 4 parts) and with many (a rig cycled per
    frame: hundreds per workgroup -> 1 part)"""
    H, W = (120, 200) if S < 20 else (72, 128)
    v = synth.make_video(S, H, W, seed=11, scene="nominal" if n_sel == "few" else "wide_baseline")
    st, cnt = ops.static_aggregate(T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], capacity=S * H * W)
    n = ops.checked_count(cnt, "agg")
    o = orc.aggregate_static_pcl(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"])
    assert n == o.shape[0]
    assert np.array_equal(N(st[:n]).view(np.uint32), o.view(np.uint32))


def test_knn_scan_with_many_tiles():
    """the look-back scan of the kNN grid with tiles taken by ticket by at most 256 looping workgroups: a volume cloud fine
    enough that the quarter-cell counters span many 8 K tiles; brute force as the checker"""
    g = torch.Generator(device="cpu").manual_seed(3)
    n = 60000
    pts = torch.rand(n, 3, generator=g) * torch.tensor([4.0, 3.0, 2.0])
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    avg = ops.knn_mean_dist(pts.to(DEV), cnt, 8)
    ref = ops.knn_mean_dist(pts.to(DEV), cnt, 8, algo=1)
    torch.cuda.synchronize()
    assert torch.equal(avg, ref)


def test_side_worker_thread_on_and_off():
    """option side_thread: the dynamic branch enqueued by the library's worker thread (default) or by the calling thread behind
    the static branch -- the same launches on the same streams: identical static images and masks, splat images to the rounding of
    the float atomics; twenty views in a row on the worker (hand-over, error path idle), the cloud aggregated inside the call"""
    from pgdvs_amd.runtime import ResidentVideoRenderer

    H, W, S = 270, 480, 6
    v = synth.make_video(S, H, W, seed=3)
    model, rc = _renderer(dyn_pcl_remove_outlier=True)
    rvr = ResidentVideoRenderer(model, rc, T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], lanes=2, side_streams=True)
    datas = [synth.to_torch(synth.make_view(v, i, frac=0.4, seed=5), DEV) for i in (1, 3)]
    rvr.calibrate(datas[0])
    assert ops.get_option("side_thread") == 1.0
    outs = {}
    for mode in (1, 0, 1):
        ops.set_option("side_thread", mode)
        try:
            rets = [rvr.render(datas[j % 2], j % 2)[0] for j in range(20)]
            rvr.join()
            torch.cuda.synchronize()
        finally:
            ops.set_option("side_thread", 1)
        for j in (0, 1, 18, 19):
            ref = outs.setdefault(j % 2, rets[j])
            assert torch.equal(rets[j]["geo_static_rgb"], ref["geo_static_rgb"]) and torch.equal(rets[j]["render_dyn_mask"], ref["render_dyn_mask"])
            assert torch.allclose(rets[j]["combined_rgb"], ref["combined_rgb"], rtol=0, atol=1e-5)
            assert int(rets[j]["geo_static_raster_status"]) == 0
