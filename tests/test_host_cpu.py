"""CPU-only checks of the host layer: the C-ABI library loads and exports exactly the
symbols include/pgdvs_hip.h declares, the product path refuses CPU tensors / a missing
library loudly, and the multi-GPU sharding + gather logic (gloo, world_size 2)."""
import os
import pathlib
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "pgdvs_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgdvs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pgdvs_amd import _lib

    assert _lib.LIB_PATH.exists(), "build the HIP library first (__graft_entry__.build())"
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.pgdvs_abi_version() == 1
    assert lib.pgdvs_build_arch() == b"gfx950"
    # pure host-side size queries work without a GPU
    assert lib.pgdvs_compact_workspace_bytes(1 << 20) >= 1024
    assert lib.pgdvs_points_raster_workspace_bytes(1000, 270, 480, 0.01) > 0
    assert lib.pgdvs_points_raster_workspace_bytes(-1, 270, 480, 0.01) == -1


def test_view_geo_description_struct_and_workspace_query():
    """the one-call-per-view entry point: this binding's struct is the library's, field for field (size check at load),
    and the pure host-side workspace query works without a GPU"""
    import ctypes as C

    from pgdvs_amd import _lib

    lib = _lib.load()
    assert lib.pgdvs_view_geo_desc_size() == C.sizeof(_lib.ViewGeoDesc)
    # the C declaration and the ctypes fields, name for name and in order
    text = (ROOT / "include" / "pgdvs_hip.h").read_text()
    body = text[text.index("typedef struct pgdvs_view_geo_desc {"):text.index("} pgdvs_view_geo_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            names += [re.sub(r"[^a-zA-Z0-9_]", "", part.split()[-1]) for part in decl.split(",")]
    assert names == [f[0] for f in _lib.ViewGeoDesc._fields_]
    d = _lib.ViewGeoDesc()
    d.H, d.W, d.agg_S, d.agg_capacity, d.row_bound, d.radius, d.K = 1080, 1920, 24, 4400000, 4400000, 0.01, 3
    d.remove_outlier, d.outlier_knn = 1, 50
    need = lib.pgdvs_view_geo_workspace_bytes(C.byref(d))
    assert 300e6 < need < 2.5e9  # (0.85 GB of it the aggregation's staging rows, 0.53 GB the rasteriser's tile segments, 0.28 GB its exact lists)
    d.remove_outlier = 0
    assert 0 < lib.pgdvs_view_geo_workspace_bytes(C.byref(d)) < need
    d.H = 0
    assert lib.pgdvs_view_geo_workspace_bytes(C.byref(d)) == -1 and b"bad H/W" in lib.pgdvs_last_error()
    assert lib.pgdvs_view_geo_forward(None, None, 0, None) == -1
    assert lib.pgdvs_eval_psnr_workspace_bytes() > 0


def test_no_cpu_fallback():
    from pgdvs_amd import ops
    from pgdvs_amd.utils.softsplat import softsplat

    x = torch.zeros(1, 3, 4, 4)
    with pytest.raises(ops.PgdvsHipError):
        softsplat(x, torch.zeros(1, 2, 4, 4), None, "sum")
    with pytest.raises(ops.PgdvsHipError):
        ops.cam_prep(torch.zeros(1, 34))


def test_product_does_not_import_oracle():
    pkg = ROOT / "ml-pgdvs_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.cpp")):
        t = f.read_text()
        assert "import oracle" not in t and "from oracle" not in t and "pgdvs_oracle" not in t, f


def test_config_surface_and_instantiate():
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.synth import DEFAULT_RENDER_CFG

    cfg = load_config(static_renderer="geo")
    rc = cfg.engine.engine_cfg.render_cfg
    assert dict(rc) == DEFAULT_RENDER_CFG  # same keys + defaults as the reference YAML
    assert cfg.model._target_ == "pgdvs_amd.renderers.pgdvs_renderer.PGDVSRenderer"
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer
    from pgdvs_amd.renderers.st_geo_renderer import StaticGeoPointRenderer

    m = PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=cfg.model.softsplat_metric_abs_alpha)
    assert isinstance(m.static_renderer, StaticGeoPointRenderer)
    assert isinstance(m, torch.nn.Module) and hasattr(m, "dyn_renderer")
    assert m.static_renderer.train(True) is m.static_renderer  # disabled_train
    cfg2 = load_config(static_renderer="gnt")
    m2 = PGDVSRenderer(cfg2, render_cfg=rc)
    assert callable(m2.static_renderer.projector.compute_projections)


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler

    from pgdvs_amd.dist import shard_indices

    for n in (1, 5, 8, 13, 288):
        for world in (1, 2, 4, 8):
            for rank in range(world):
                ref = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False))
                assert shard_indices(n, rank, world) == ref, (n, world, rank)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "ml-pgdvs_amd"))
from pgdvs_amd.dist import shard_indices, gather_image_stack, reduce_metrics, AsyncImageGather
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_views = 5
mine = shard_indices(n_views, rank, world)
local = torch.stack([torch.full((3, 4, 6), float(v)) for v in mine])
out = gather_image_stack(local, n_views)
m = reduce_metrics(torch.tensor([1.0, float(rank)]))
for kw in ({}, dict(n_steps=3, like=torch.empty(1, 3, 2, 2)), dict(n_steps=2, like=torch.empty(1, 3, 2, 2))):
    ag = AsyncImageGather(**kw)  # per-step buffers / preallocated stack / stack too short for the third step
    for step in range(3):
        ag.submit(torch.full((1, 3, 2, 2), float(step * world + rank)))
    stack = ag.finish()
    if rank == 0:
        assert stack.shape == (3 * world, 3, 2, 2), stack.shape
        assert [int(stack[i, 0, 0, 0]) for i in range(3 * world)] == list(range(3 * world)), stack[:, 0, 0, 0]
        assert (ag.stack is not None) == bool(kw)
    else:
        assert stack is None
# bounded receive memory: a ring of 3 slots serves 8 steps; every view is consumed (checksummed) before its slot is
# reused, rank 0 holds 3 x world images instead of 8 x world, and the last ring-full comes back in view order
ag = AsyncImageGather(n_steps=8, like=torch.empty(1, 3, 2, 2), ring=3)
assert ag.local.shape[0] == 3 and (ag.stack is None or ag.stack.shape[:2] == (3, world))
seen = []
for step in range(8):
    s = ag.slot()
    s.fill_(float(step * world + rank))
    ag.submit(s)
res = ag.finish()
if rank == 0:
    assert res["sums"].shape == (8, world)
    assert res["sums"].flatten().tolist() == [12.0 * v for v in range(8 * world)], res["sums"]
    assert res["tail_first_step"] == 5 and [int(x) for x in res["tail"][:, 0, 0, 0]] == list(range(5 * world, 8 * world))
else:
    assert res is None
# the same buffers serve a second, shorter run (reset): nothing is allocated between runs
ptr = ag.local.data_ptr()
ag.reset(4)
for step in range(4):
    ag.submit(torch.full((1, 3, 2, 2), float(100 + step * world + rank)))
res2 = ag.finish()
assert ag.local.data_ptr() == ptr
if rank == 0:
    assert res2["sums"].flatten().tolist() == [12.0 * (100 + v) for v in range(4 * world)], res2["sums"]
    assert res["sums"].flatten().tolist() == [12.0 * v for v in range(8 * world)]  # the first run's result is its own copy
# a caller-supplied consumer sees every step exactly once, in order, with all ranks' images
got = []
ag = AsyncImageGather(n_steps=5, like=torch.empty(1, 3, 2, 2), ring=2, consumer=lambda i, imgs: got.append((i, imgs[:, 0, 0, 0, 0].tolist())))
for step in range(5):
    ag.submit(torch.full((1, 3, 2, 2), float(step * world + rank)))
ag.finish()
if rank == 0:
    assert got == [(i, [float(i * world + r) for r in range(world)]) for i in range(5)], got
else:
    assert got == []
# evaluator-shaped step on every rank: the metric sums arrive on rank 0 through ONE packed reduce
from pgdvs_amd.harness import eval_step
class Fake(torch.nn.Module):
    def forward(self, data_gpu, render_cfg=None, disable_tqdm=True, for_debug=False):
        return {"combined_rgb": torch.full((2, 3, 4, 6), 0.25 * (rank + 1))}
d = {"rgb_src_temporal": torch.zeros(2, 2, 4, 6, 3), "rgb_tgt": torch.full((2, 4, 6, 3), 0.5), "eval_mask": torch.zeros(2, 4, 6, 3)}
md = eval_step(Fake(), d, None, device="cpu")
if rank == 0:
    import math
    q = lambda x: float((torch.tensor(x) * 255).byte().float() / 255.0)
    # (rank 1 predicts the ground truth exactly: the reference's PSNR is 0 for identical images)
    want = sum(2 * 10 * math.log10(1.0 / (q(0.5) - q(0.25 * (r + 1))) ** 2) for r in range(world) if q(0.5) != q(0.25 * (r + 1)))
    assert int(md["eval/count"]) == 2 * world and abs(float(md["eval/psnr_full_combined"]) - want) < 1e-3, (md, want)
    assert float(md["eval/psnr_dyn_combined"]) == 0.0
if rank == 0:
    assert out.shape == (n_views, 3, 4, 6), out.shape
    assert [int(out[i, 0, 0, 0]) for i in range(n_views)] == list(range(n_views))
    assert m.tolist() == [2.0, 1.0]
    print("GATHER_OK")
else:
    assert out is None
dist.destroy_process_group()
"""


def test_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29713", str(script), str(ROOT)],
        capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK" in r.stdout


def _bench(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=env)


def test_bench_gpus_2_starts_two_ranks_dry():
    """`python bench.py --gpus 2` (the driver's form, no launcher around it) must start two ranks by
    itself, shard the views, gather them to rank 0 and relay ONE JSON line (reference: run.py:158-176)."""
    import json

    r = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["views_gathered"] == 8 and len(out["per_rank_seconds"]) == 2
    assert out["gather_bytes_to_rank0"] == 4 * 3 * 4 * 6 * 4


def test_bench_launcher_reports_a_failed_rank():
    r = _bench("--gpus", "2", "--steps", "2", "--dry-run", "--dry-fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_watchdog_ends_a_job_with_a_lost_rank():
    """a rank that never reaches its peers again: the other rank's wall-clock timeout ends ITS process with a non-zero
    status (fresh-process semantics: it exits, nothing is re-executed), the launcher tears the job down and returns
    non-zero -- instead of hanging inside a collective"""
    import time

    t0 = time.time()
    r = _bench("--gpus", "2", "--steps", "2", "--dry-run", "--dry-hang-rank", "1", "--rank-timeout", "6")
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert "did not finish within 6 s" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_gpus_8_dry_run_c4_sharding_and_ring():
    """first contact with an 8-GPU node can then only find RCCL itself: `python bench.py --gpus 8` as the driver starts it,
    BASELINE.json configs[3]'s 288 target views as 36 per rank (reference: trainer_pgdvs.py:290-306, run.py:158-176), on CPU
    tensors over gloo: eight ranks come up, every view arrives on rank 0 exactly once and in view order through a receive
    ring of the size the real run uses with its fixed four lanes (12 slots x 8 ranks, not 36 x 8), per-rank times come back"""
    import json

    sys.path.insert(0, str(ROOT))
    import bench

    r = _bench("--gpus", "8", "--steps", "36", "--warmup", "2", "--dry-run")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["value"] is None and out["scaling"] == "weak"
    assert out["views_gathered"] == 288 and len(out["per_rank_seconds"]) == 8
    assert out["receive_ring_slots"] == bench.ring_slots_for(6, bench.DEFAULT_LANES_MULTI_RANK) == 12 and bench.DEFAULT_ARRANGEMENT_MULTI_RANK == (4, False, True)
    assert out["gather_bytes_to_rank0"] == 36 * 7 * 3 * 4 * 6 * 4
    # the reference's sampler over the same 288 views: rank r renders r, r + 8, ... (what `(j * world + rank)` enumerates)
    from pgdvs_amd.dist import shard_indices

    assert all(shard_indices(288, rk, 8) == [j * 8 + rk for j in range(36)] for rk in range(8))
    # ... and a count that does not divide: the tail wraps around to the first views (DistributedSampler's padding)
    assert shard_indices(285, 5, 8)[-1] == (35 * 8 + 5) % 285 == 0


def test_bench_watchdog_with_rank_5_of_8_missing():
    """eight ranks, rank 5 never reaches the timed loop: the seven others leave through their own wall-clock limit (status
    124 each, nothing re-executed), the launcher returns non-zero and prints no line"""
    import time

    t0 = time.time()
    r = _bench("--gpus", "8", "--steps", "4", "--dry-run", "--dry-hang-rank", "5", "--rank-timeout", "8")
    assert r.returncode != 0
    assert time.time() - t0 < 200
    assert "did not finish within 8 s" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_library_options_roundtrip_without_a_gpu():
    """include/pgdvs_hip.h "Options": process-wide switches read from the environment once, at load time, and changed only
    through pgdvs_option_set afterwards (no entry point calls getenv); defaults, set / get, unknown names"""
    import ctypes as C
    import math

    from pgdvs_amd import _lib

    lib = _lib.load()
    defaults = {"agg_ordered": 0.0, "agg_stage": 1.0, "gnt_fp32": 0.0, "knn_no_tpq": 0.0, "knn_stats": 0.0}
    for k, v in defaults.items():
        if ("PGDVS_" + k.upper()) not in os.environ:
            assert lib.pgdvs_option_get(k.encode()) == v, k
    assert lib.pgdvs_option_get(b"raster_bound_density") == pytest.approx(float(os.environ.get("PGDVS_RASTER_BOUND_DENSITY", 2.2)))
    prev = lib.pgdvs_option_get(b"gnt_fp32")
    assert lib.pgdvs_option_set(b"gnt_fp32", C.c_double(7.0)) == 0 and lib.pgdvs_option_get(b"gnt_fp32") == 1.0  # flags: value != 0
    assert lib.pgdvs_option_set(b"gnt_fp32", C.c_double(prev)) == 0
    assert lib.pgdvs_option_set(b"raster_bound_density", C.c_double(1.25)) == 0
    assert lib.pgdvs_option_get(b"raster_bound_density") == 1.25
    assert lib.pgdvs_option_set(b"raster_bound_density", C.c_double(2.2)) == 0
    assert lib.pgdvs_option_set(b"no_such_option", C.c_double(1.0)) < 0 and b"no_such_option" in lib.pgdvs_last_error()
    assert math.isnan(lib.pgdvs_option_get(b"no_such_option"))
    # a later setenv changes nothing: the environment was read when the library was loaded
    os.environ["PGDVS_GNT_FP32"] = "1"
    try:
        assert lib.pgdvs_option_get(b"gnt_fp32") == prev
    finally:
        os.environ.pop("PGDVS_GNT_FP32", None)
    src = "".join((ROOT / "ml-pgdvs_amd" / "csrc" / f).read_text() for f in os.listdir(ROOT / "ml-pgdvs_amd" / "csrc")
                  if f.endswith((".hip", ".cpp", ".h")) and f != "error.cpp")
    code = re.sub(r"//[^\n]*", "", src)
    code = "\n".join(ln for ln in code.splitlines() if "PGDVS_AB_CHAIN" not in ln and "PGDVS_DBG_CHAIN" not in ln)
    assert code.count("getenv(") <= 1, "an entry point reads the environment (only error.cpp and the A/B switch of static_agg.hip may)"


def test_bench_rejects_world_size_mismatch():
    r = _bench("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr


def test_gnt_modules_match_reference_golden(golden_dir):
    """Host-side GNT modules (state-dict layout, seeded init order, dense mask-driven
    formulation of the view/ray transformers) against vectors produced by the reference."""
    import numpy as np

    from pgdvs_amd.models.gnt.model import GNTModel

    g = dict(np.load(golden_dir / "gnt_small.npz"))
    r = dict(np.load(golden_dir / "gnt_resunet.npz"))
    torch.manual_seed(int(r["seed"]))
    m = GNTModel(netwidth=64, transformer_depth=2).eval()
    assert sum(p.numel() for p in m.feature_net.parameters()) == int(r["n_params"])
    with torch.no_grad():
        f = m.feature_net(torch.from_numpy(r["src_rgbs"][0]).permute(0, 3, 1, 2))[0].numpy()
    np.testing.assert_allclose(f, r["feat"], rtol=0, atol=1e-5)  # same init sequence as the reference
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w_")}
    m.net_coarse.load_state_dict(sd, strict=True)  # checkpoint-compatible parameter names
    T = torch.from_numpy
    for tag in ("nomask", "dynmask"):
        with torch.no_grad():
            out, ex = m.net_coarse(T(g[f"{tag}_rgb_feat"]), T(g[f"{tag}_ray_diff"]), T(g[f"{tag}_mask"]), T(g["pts"]),
                                   T(g["ray_d"]), ret_view_entropy=True, ret_view_std=True)
        np.testing.assert_allclose(out.numpy(), g[f"{tag}_out"], rtol=0, atol=1e-5)
        for k, v in ex.items():
            np.testing.assert_allclose(v.numpy(), g[f"{tag}_{k}"], rtol=0, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("case", ["v10", "v24"])
def test_gnt_mirror_matches_reference_at_depth_8(golden_dir, case):
    """the torch mirror (CPU branch: the dense mask-driven formulation) against the reference's own forward at its configured
    depth -- 8 layers, 256 samples per ray, 10 / 24 source views (gnt_depth8.npz; transformer_network.py:423-539) -- incl.
    the hidden state behind each of the 16 blocks (subsampled by the fixture)"""
    import sys

    import numpy as np

    sys.path.insert(0, str(golden_dir))
    import gnt_depth8_inputs as GI

    from pgdvs_amd.models.gnt.models.transformer_network import GNT

    g = dict(np.load(golden_dir / "gnt_depth8.npz"))
    net = GNT(netwidth=64, transformer_depth=8).eval()
    w = GI.make_weights({k: tuple(v.shape) for k, v in net.state_dict().items()})
    assert abs(GI.checksum(w) - float(g["weights_checksum"])) <= 1e-9 * abs(float(g["weights_checksum"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    x = {k: torch.from_numpy(v) for k, v in GI.make_inputs(case).items()}
    hidden = []
    net.hidden_hook = lambda name, q: hidden.append(q[:, ::16, ::4].clone())
    with torch.no_grad():
        out, ex = net(x["rgb_feat"], x["ray_diff"], x["mask"], x["pts"], x["ray_d"], ret_view_entropy=True, ret_view_std=True)
    np.testing.assert_allclose(out.numpy(), g[f"{case}_out"], rtol=0, atol=2e-5)
    for k, v in ex.items():
        np.testing.assert_allclose(v.numpy(), g[f"{case}_{k}"], rtol=0, atol=2e-5, err_msg=k)
    assert len(hidden) == 16
    ref = g[f"{case}_hidden"]
    for i, h in enumerate(hidden):
        np.testing.assert_allclose(h.numpy(), ref[i], rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref[i]).max())), err_msg=f"block {i}")


def test_harness_quantisation_and_psnr_vs_reference(golden_dir):
    import numpy as np

    from pgdvs_amd.harness import masked_psnr, quantize_like_evaluator

    g = dict(np.load(golden_dir / "harness_psnr.npz"))
    pq = quantize_like_evaluator(torch.from_numpy(g["pred"]))
    gq = quantize_like_evaluator(torch.from_numpy(g["gt"]))
    assert np.array_equal(pq.numpy(), g["pred_q"]) and np.array_equal(gq.numpy(), g["gt_q"])
    m = torch.from_numpy(g["mask"])
    assert abs(masked_psnr(pq, gq, m) - float(g["psnr"])) < 1e-9
    assert masked_psnr(gq, gq, m) == float(g["psnr_same"]) == 0


def test_harness_eval_step_vs_reference_eval_step(golden_dir):
    """pgdvs_amd.harness.eval_step against vectors produced by the reference's own PGDVSEvaluator.eval_step
    (tests/golden/make_golden_harness.py) around a stand-in model: same clamp / NaN / quantisation / resize of
    the ground truth / masked PSNR sums and count"""
    import numpy as np

    from pgdvs_amd.harness import METRIC_KEYS, eval_step

    g = dict(np.load(golden_dir / "harness_eval_step.npz"))
    for tag in ("same", "strided"):
        pred = torch.from_numpy(g[f"{tag}_pred"])

        class Fake(torch.nn.Module):
            def forward(self, data_gpu, render_cfg=None, disable_tqdm=True, for_debug=False):
                assert render_cfg == "rc" and for_debug is False
                return {"combined_rgb": pred}

        B, H, W, _ = g[f"{tag}_gt"].shape
        data = {"rgb_src_temporal": torch.zeros(B, 2, H, W, 3), "rgb_tgt": torch.from_numpy(g[f"{tag}_gt"]),
                "eval_mask": torch.from_numpy(g[f"{tag}_mask"]), "misc": [{}] * B}
        md = eval_step(Fake(), data, "rc", device="cpu")
        assert int(md["eval/count"]) == int(g[f"{tag}_metric__eval__count"][0]) == B
        for k in METRIC_KEYS:
            assert md[f"eval/{k}"].dtype == torch.float32
            np.testing.assert_allclose(float(md[f"eval/{k}"]), float(g[f"{tag}_metric__eval__{k}"]), rtol=1e-6, err_msg=f"{tag} {k}")


# ---------------------------------------------------------------- on-disk formats -> data dict (8f-3)
def _digest(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a @ np.random.default_rng(12345).random(a.size), a.sum(), a.min(), a.max()])


@pytest.fixture(scope="module")
def nvidia_tree(tmp_path_factory):
    import sys

    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent / "golden"))
    import nvidia_tree as NT

    return NT, NT.build_tree(tmp_path_factory.mktemp("nvidia"))


def test_nvidia_dataset_items_vs_reference(golden_dir, nvidia_tree):
    """the mirror dataset, pointed at a rebuilt copy of the synthetic tree, returns what the
    reference's NvidiaDynEvaluationDataset returned on it (tests/golden/make_golden_nvidia.py)"""
    from pgdvs_amd.datasets.nvidia_eval import NvidiaDynEvaluationDataset, read_llff_cams

    NT, root = nvidia_tree
    g = dict(np.load(golden_dir / "nvidia_items.npz"))
    hwf, c2w = read_llff_cams(root / "raw" / NT.SCENE / "dense" / "poses_bounds_cvd.npy")
    assert np.array_equal(hwf, g["cam_hwf"]) and np.array_equal(c2w, g["cam_c2w"])
    ds = NvidiaDynEvaluationDataset(
        data_root=root, raw_data_dir="raw", depth_data_dir="depths", mask_data_dir="masks", flow_data_dir="flows", max_hw=-1,
        mode="eval", scene_ids=[NT.SCENE], n_src_views_spatial=4, n_src_views_temporal_track_one_side=2, flow_consist_thres=1.0)
    assert len(ds) == NT.F * NT.N_CAMS
    for n, (f, c) in enumerate(g["items"]):
        item = ds[int(f) * NT.N_CAMS + int(c)]
        assert item["misc"] == {"scene_id": NT.SCENE, "tgt_frame_id": int(f), "tgt_cam_id": int(c)}
        ref_keys = {k[len(f"i{n}_"):].split("__")[0] for k in g if k.startswith(f"i{n}_")}
        derived = {k for k in item if k.startswith("dyn_rgb") or k.startswith("static_rgb")}
        assert set(item.keys()) - {"scene_id", "misc"} - derived == ref_keys
        for k in sorted(ref_keys):
            v = item[k].numpy()
            if k.startswith("rgb_"):
                v = np.round(v * 255.0)
            if f"i{n}_{k}" in g:
                ref = g[f"i{n}_{k}"]
                assert v.shape == ref.shape, k
                if np.issubdtype(ref.dtype, np.integer):
                    assert np.array_equal(v, ref), k  # frame selections, counts
                else:
                    np.testing.assert_allclose(v, ref, rtol=1e-6, atol=1e-7, err_msg=k)  # cameras, times, depth_range
            else:
                assert tuple(v.shape) == tuple(g[f"i{n}_{k}__shape"]), k
                np.testing.assert_allclose(_digest(v), g[f"i{n}_{k}__digest"], rtol=1e-7, atol=1e-9, err_msg=k)
        for sfx in ("spatial", "temporal", "temporal_track_fwd2tgt", "temporal_track_bwd2tgt"):
            m = item[f"dyn_mask_src_{sfx}"]
            assert torch.equal(item[f"dyn_rgb_src_{sfx}"], item[f"rgb_src_{sfx}"] * m)
            assert torch.equal(item[f"static_rgb_src_{sfx}"], item[f"rgb_src_{sfx}"] * (1 - m))


def test_mono_dataset_items_vs_reference(golden_dir, tmp_path):
    """in-the-wild video layout -> data dict along the bullet-time path: the mirror returns what the
    reference's MonoVisualizationDataset returned on the same synthetic tree"""
    import sys

    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent / "golden"))
    import nvidia_tree as NT
    from pgdvs_amd.datasets.mono_vis import MonoVisualizationDataset

    NT.build_mono_tree(tmp_path)
    g = dict(np.load(golden_dir / "mono_items.npz"))
    ds = MonoVisualizationDataset(
        data_root=tmp_path, max_hw=-1, mode="vis", scene_ids=[NT.MONO_SCENE], n_src_views_spatial=3,
        n_src_views_temporal_track_one_side=2, vis_center_time=4, n_render_frames=16, vis_time_interval=3, vis_bt_max_disp=8,
        flow_consist_thres=1.0)
    assert len(ds) == int(g["n_items"])
    np.testing.assert_allclose(np.array([e[2] for e in ds.valid_fs]), g["all_tgt_time"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.stack([e[4] for e in ds.valid_fs]), g["all_tgt_c2w"], rtol=1e-9, atol=1e-10)
    for n, idx in enumerate(g["items"]):
        item = ds[int(idx)]
        assert item["misc"]["scene_id"] == NT.MONO_SCENE and item["misc"]["tgt_idx"] == int(idx)
        ref_keys = {k[len(f"i{n}_"):].split("__")[0] for k in g if k.startswith(f"i{n}_")}
        derived = {k for k in item if k.startswith("dyn_rgb") or k.startswith("static_rgb")}
        assert set(item.keys()) - {"scene_id", "misc"} - derived == ref_keys
        for k in sorted(ref_keys):
            v = item[k].numpy()
            if k.startswith("rgb_"):
                v = np.round(v * 255.0)
            if f"i{n}_{k}" in g:
                ref = g[f"i{n}_{k}"]
                assert v.shape == ref.shape, k
                if np.issubdtype(ref.dtype, np.integer):
                    assert np.array_equal(v, ref), k
                else:
                    np.testing.assert_allclose(v, ref, rtol=1e-6, atol=1e-7, err_msg=k)
            else:
                assert tuple(v.shape) == tuple(g[f"i{n}_{k}__shape"]), k
                np.testing.assert_allclose(_digest(v), g[f"i{n}_{k}__digest"], rtol=1e-7, atol=1e-9, err_msg=k)
        for sfx in ("spatial", "temporal", "temporal_track_fwd2tgt", "temporal_track_bwd2tgt"):
            m = item[f"dyn_mask_src_{sfx}"]
            assert torch.equal(item[f"dyn_rgb_src_{sfx}"], item[f"rgb_src_{sfx}"] * m)
            assert torch.equal(item[f"static_rgb_src_{sfx}"], item[f"rgb_src_{sfx}"] * (1 - m))


def test_config_surface_groups_and_combined_dataset(nvidia_tree):
    """every group of the reference's config tree exists in the mirror (configs/pgdvs.yaml defaults list,
    engine/{evaluator,visualizer}_pgdvs, dataset/combined, tracker/{dummy,tapnet,cotracker}); the dataset group
    instantiates the mirror's CombinedDataset over the mirrored loaders (configs/dataset/combined.yaml)"""
    from pgdvs_amd.datasets.combined import CombinedDataset, dataset_class
    from pgdvs_amd.datasets.nvidia_eval import NvidiaDynEvaluationDataset
    from pgdvs_amd.instantiate import instantiate, load_config

    cfg = load_config()
    assert cfg.static_renderer._target_ == "pgdvs_amd.models.gnt.renderer.BaseRenderer" and cfg.tracker._target_ is None
    assert cfg.engine._target_ == "pgdvs.engines.evaluator_pgdvs.PGDVSEvaluator"  # engines stay the reference's
    assert load_config(engine="visualizer_pgdvs").engine.engine_cfg.render_cfg == cfg.engine.engine_cfg.render_cfg
    assert load_config(tracker="tapnet").tracker.query_chunk_size == load_config(tracker="cotracker").tracker.query_chunk_size == 4096
    ds_cfg = cfg.dataset
    assert ds_cfg._target_ == "pgdvs_amd.datasets.combined.CombinedDataset"
    assert ds_cfg.max_hw == -1 and ds_cfg.rgb_range == "0_1"                      # ${dataset_max_hw}, ${rgb_range}
    assert ds_cfg.dataset_specifics.nvidia_vis.vis_bt_max_disp == 64              # ${vis_specifics.vis_bt_max_disp}
    assert dataset_class("nvidia_eval") is NvidiaDynEvaluationDataset
    with pytest.raises(KeyError):
        dataset_class("no_such_dataset")
    NT, root = nvidia_tree
    spec = dict(ds_cfg.dataset_specifics.nvidia_eval)
    spec.update(scene_ids=[NT.SCENE], raw_data_dir="raw", depth_data_dir="depths", mask_data_dir="masks", flow_data_dir="flows",
                n_src_views_spatial=4, n_src_views_temporal_track_one_side=2)
    spec.pop("use_zoe_depth"), spec.pop("zoe_depth_data_path")
    node = dict(ds_cfg)
    node.update(data_root=root, dataset_specifics={"nvidia_eval": spec})
    ds = instantiate(node, mode="eval")
    assert isinstance(ds, CombinedDataset) and len(ds) == NT.F * NT.N_CAMS
    direct = ds.datasets["nvidia_eval"][7]
    item = ds[7]
    assert item["misc"] == direct["misc"] and torch.equal(item["flat_cam_tgt"], direct["flat_cam_tgt"])
    with pytest.raises(IndexError):
        ds[len(ds)]


def test_nvidia_frame_selection_rules():
    from pgdvs_amd.datasets.nvidia_eval import select_temporal_frames

    s = select_temporal_frames(5, 5, 14, 2)      # inside the mono video: both neighbours
    assert s["temporal"] == [4, 6] and s["n_actual_temporal"] == 2 and s["fwd2tgt"] == [2, 3] and s["bwd2tgt"] == [7, 8]
    s = select_temporal_frames(0, 0, 14, 2)      # first frame: one neighbour, duplicated placeholder
    assert s["temporal"] == [1, 1] and s["n_actual_temporal"] == 1 and s["n_actual_fwd2tgt"] == 0 and s["bwd2tgt"] == [2, 3]
    s = select_temporal_frames(6, 2, 14, 2)      # another camera at time 6: the input frame of that instant
    assert s["temporal"] == [6, 6] and s["n_actual_temporal"] == 1 and s["fwd2tgt"] == [4, 5] and s["bwd2tgt"] == [7, 8]


def test_gnt_fine_sampling_mirror_vs_torch_reference_formula():
    """sample_pdf / sample_fine_z: deterministic inverse-CDF sampling; checked against a direct
    numpy evaluation of the same piecewise-linear inverse (the end-to-end pin against the
    reference's BaseRenderer fine pass is the GPU test on gnt_render.npz)"""
    from pgdvs_amd.models.gnt.ray_sampler import sample_pdf, sample_fine_z

    rng = np.random.default_rng(3)
    R, M, Ns = 7, 10, 6
    bins = np.sort(rng.uniform(1.0, 4.0, (R, M + 1)), axis=1).astype(np.float32)
    w = rng.random((R, M)).astype(np.float32)
    s = sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), Ns, det=True).numpy()
    pdf = (w + 1e-5) / (w + 1e-5).sum(1, keepdims=True)
    cdf = np.concatenate([np.zeros((R, 1), np.float32), np.cumsum(pdf, 1)], 1)
    u = np.linspace(0, 1, Ns, dtype=np.float32)
    for r in range(R):
        for j in range(Ns):
            above = int((u[j] >= cdf[r, :M]).sum())
            below = max(above - 1, 0)
            den = cdf[r, above] - cdf[r, below]
            den = 1.0 if den < 1e-5 else den
            exp = bins[r, below] + (u[j] - cdf[r, below]) / den * (bins[r, above] - bins[r, below])
            assert abs(s[r, j] - exp) < 1e-5
    z = np.sort(rng.uniform(1.0, 4.0, (R, M + 2)), axis=1).astype(np.float32)
    zz = sample_fine_z(True, Ns, True, torch.from_numpy(rng.random((R, M + 2)).astype(np.float32)), torch.from_numpy(z)).numpy()
    assert zz.shape == (R, M + 2 + Ns) and np.all(np.diff(zz, axis=1) >= 0)
    assert np.all(zz.min(1) >= z.min(1) - 1e-6) and np.all(zz.max(1) <= z.max(1) + 1e-6)


def test_bench_compact_line_from_a_full_record_stays_under_4k():
    """BENCH_r05.json had `parsed: null`: the line had grown to 22.5 KB and the driver's bounded read cut its head off.  The
    line now goes through bench.compact_line; round 5's full record (profiles/r05_bench_line.json) reduced by the same function
    must fit in 4 KB, keep the contract's keys with `roofline` and `cpu_baseline`, and carry no long prose"""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.loads((ROOT / "profiles" / "r05_bench_line.json").read_text())
    assert len(json.dumps(full)) > 20000
    s = bench.compact_line(full)
    assert len(s.encode()) <= 4096 and "\n" not in s
    b = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["value"] == full["value"] and b["config"]["workload"].startswith("1920x1080")
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in b["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in b["cpu_baseline"], k
    assert "kernels" not in b and "variants" not in b and "roofline_kernels" not in b

    def strings(o):
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)

    assert max(len(x) for x in strings(b)) <= 200
    # a worst case: 8 ranks, every optional object present, long strings everywhere
    full["config"]["rccl_ranks_seen"] = list(range(8))
    full["config"]["per_rank_frames_per_s"] = [1234.56] * 8
    full["config"]["launch"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    assert len(bench.compact_line(full).encode()) <= 4096
