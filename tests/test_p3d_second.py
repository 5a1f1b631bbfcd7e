"""Row A9's unpinned boundary (pytorch3d 0.7.4, un-vendored and unobtainable offline): the closed-form
oracle (oracle/pgdvs_oracle.c) against an independent second restatement that follows pytorch3d's own
object chain (oracle/p3d_second.py: cameras_from_opencv_projection -> Rotate.compose(Translate) ->
Transform3d.transform_points -> K^T -> RasterizePointsNaiveCpu's priority queue -> NormWeightedCompositor),
and a measurement of how much the float32 accumulation order that reading cannot settle (sequential vs
fused multiply-add inside torch.bmm, i.e. pytorch3d's CPU vs CUDA backends) changes the z-buffer index.
CPU only."""
import sys

import numpy as np

from oracle import oracle as orc
from oracle import p3d_second as p3d
from pgdvs_amd import synth


def _small_scene(seed, H=24, W=40, n=500):
    rng = np.random.default_rng(seed)
    K3, c2w = synth.frame_camera(3, 8, H, W)
    fc = synth.flat_cam(H, W, K3, c2w)
    u, v = rng.uniform(-3, W + 3, n), rng.uniform(-3, H + 3, n)
    z = np.round(rng.uniform(1.5, 3.0, n), 1)  # coarse depths: many exact z ties (broken by index)
    z[rng.integers(0, n, 20)] = -0.5              # behind the camera: skipped
    cam = np.stack([(u - K3[0, 2]) / K3[0, 0] * z, (v - K3[1, 2]) / K3[1, 1] * z, z], 1)
    world = (cam @ c2w[:3, :3].T + c2w[:3, 3]).astype(np.float32)
    return fc, world, rng.random((n, 3)).astype(np.float32)


def test_second_restatement_matches_closed_form_oracle_bit_for_bit():
    """same float32 accumulation ("seq"): the object-by-object chain and the closed form give identical NDC
    coordinates; the priority-queue rasteriser and the C rasteriser give identical idx / zbuf / dist2
    (z ties and points behind the camera included); composites agree to rounding"""
    for seed, (H, W), K, radius in [(0, (24, 40), 3, 0.08), (1, (40, 24), 1, 0.05), (2, (32, 32), 8, 0.15)]:
        fc, world, feat = _small_scene(seed, H, W)
        ndc = p3d.points_to_ndc(fc, world, flavour="seq", inverse="f64")
        assert np.array_equal(ndc.view(np.uint32), orc.points_to_ndc(world, fc, H, W).view(np.uint32))
        idx, zbuf, d2 = p3d.rasterize_points_naive(ndc, H, W, radius, K)
        o_idx, o_z, o_d2 = orc.rasterize_points(world, fc, H, W, radius, K)
        assert (idx >= 0).mean() > 0.3
        assert np.array_equal(idx, o_idx)
        assert np.array_equal(zbuf.view(np.uint32), o_z.view(np.uint32)) and np.array_equal(d2.view(np.uint32), o_d2.view(np.uint32))
        img = p3d.norm_weighted_composite(idx, d2, radius, feat)
        np.testing.assert_allclose(img, orc.composite(o_idx, o_d2, radius, feat), rtol=0, atol=1e-6)
        # the LAPACK float32 inverse torch.inverse uses on the CPU: same camera to rounding
        ndc32 = p3d.points_to_ndc(fc, world, flavour="seq", inverse="f32")
        np.testing.assert_allclose(ndc32, ndc, rtol=0, atol=2e-6)


def test_pix_to_non_square_ndc_matches_oracle_raster_grid():
    """pixel centres: reversed index, +0.5, longer side scaled by the aspect ratio"""
    for H, W in [(24, 40), (40, 24), (32, 32), (1080, 1920)]:
        s = min(H, W) / 2.0
        for i in (0, 1, W // 2, W - 1):
            x = p3d.pix_to_non_square_ndc(W - 1 - i, W, H)
            assert abs(float(x) - (-(i + 0.5 - W / 2.0) / s)) < 2e-6  # = -(pixel centre - principal point) / s
        assert float(p3d.non_square_ndc_range(W, H)) == float(np.float32(2.0 * W) / np.float32(H) if W > H else 2.0)


def test_backend_accumulation_order_sensitivity_is_small_and_measured():
    """pytorch3d's own backends (sequential vs fused multiply-add in torch.bmm) on 150 k points with edge-grazing
    discs and exact z ties: ~40 % of the NDC coordinates change in the last bit, the z-buffer index on a few
    1e-4 of the entries at most.  That is the size of the hole `parity unpinned` leaves for A9; the full
    10^6-point run is profiles/r02_p3d_order_sensitivity.json (tests/p3d_order_sensitivity.py)."""
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent))
    import p3d_order_sensitivity as S

    r = S.measure(n=150_000, win=32)
    f = r["flavours"]
    assert f["seq_f64inv"]["idx_entries_differ"] == 0 and f["seq_f64inv"]["ndc_xy_bits_differ_frac"] == 0.0
    assert 0.05 < f["fma_f64inv"]["ndc_xy_bits_differ_frac"] < 0.8          # the flavours really differ ...
    assert f["fma_f64inv"]["max_abs_ndc_xy_diff"] < 1e-6                      # ... by an ulp or two
    assert f["fma_f32inv"]["idx_entries_differ_frac"] < 2e-3                  # and rarely change a decision
    print("A9 order sensitivity:", {k: (v["idx_entries_differ"], r["entries_compared"]) for k, v in f.items()})
