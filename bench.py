#!/usr/bin/env python3
"""Headline benchmark: novel-view frames/s of the PGDVS rendering hot path on MI355X.

A *step* renders one novel view at H x W from S source frames that are already resident
in HBM: static point-cloud aggregation over the S frames (A12), point z-buffer
rasterisation + compositing of that cloud into the target view (A9), the dynamic branch
(unproject + flow warp + kNN outlier filter + projection, A1-A5), softmax splatting with
its metric (A6-A8) and the static/dynamic composite (A11).  Nothing is cached between
steps.  Default workload = BASELINE.json configs[2]: 1080p, 24 source frames.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line
on rank 0; for N > 1 it is launched through torch.distributed.run (one rank per GPU, RCCL),
frames are sharded across ranks (weak scaling: K views per rank) and the final image
stacks are gathered to rank 0 inside the timed region.
"""
import argparse
import contextlib
import ctypes
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "ml-pgdvs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


LINE_LIMIT = 4096  # bytes of the final stdout line (the driver reads a bounded tail of stdout: round 5's 22.5 KB line was cut)


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def compact_line(out):
    """the ONE line the driver parses, reduced from the full record `out`: the contract's keys, `roofline` and
    `cpu_baseline` with their required fields, the whole path's fraction and five scalars -- at most LINE_LIMIT bytes,
    no string longer than 200 characters.  Everything else (per-kernel table, co-dominant kernels, scene / configuration
    variants, the GNT object, prose) is the detail record: stderr + gpurun_out/bench_detail.json."""
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    c = _pick(cfg, ("workload", "height", "width", "src_frames", "scene", "views_in_flight", "launch", "parallelism",
                    "rccl_ranks_seen", "stream_queue_groups", "stream_placement_per_rank", "per_rank_frames_per_s", "gather_bytes_to_rank0",
                    "whole_view_alg_bytes")) or {}
    for k in ("workload", "launch", "parallelism"):
        if k in c:
            c[k] = _clip(c[k], 200)
    line["config"] = c
    r = _pick(out.get("roofline"), ("kernel", "bound", "alg_bytes", "avg_ms", "launches_per_view", "achieved", "peak", "unit", "frac",
                                    "traffic", "traffic_source"))
    if r and isinstance((out.get("roofline") or {}).get("rocprofv3"), dict):
        r["rocprofv3"] = _pick(out["roofline"]["rocprofv3"], ("one_view_avg_us", "one_view_frac", "three_lanes_avg_us", "three_lanes_frac"))
    if r:
        r["alg_bytes"] = int(round(r["alg_bytes"])) if r.get("alg_bytes") is not None else None
        r["traffic_source"] = _clip(r.get("traffic_source"), 120)
    line["roofline"] = r
    line["roofline_path"] = _pick(out.get("roofline_path"), ("bound", "achieved", "peak", "unit", "frac", "alg_bytes_per_view", "traffic_bytes_per_view"))
    cb = _pick(out.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample", "estimated_seconds_per_view"))
    if cb:
        cb["sample"] = _clip(cb.get("sample"), 200)
    line["cpu_baseline"] = cb
    line["steady_state"] = _pick(out.get("steady_state"), ("frames_per_s", "steps"))
    line["latency_ms"] = _pick(out.get("latency_ms"), ("median",))
    line["eval_step_frames_per_s"] = out.get("eval_step_frames_per_s")
    line["gnt"] = _pick(out.get("gnt"), ("tflops", "fp32_equivalent_frac"))
    if out.get("dry_run"):
        line["dry_run"] = True
    line["detail"] = "gpurun_out/bench_detail.json (also on stderr)"
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_LIMIT and "\n" not in s, f"bench line is {len(s)} bytes (> {LINE_LIMIT})"
    return s


def emit(out):
    """full record -> stderr and gpurun_out/bench_detail.json; the compact line -> the LAST thing on stdout"""
    detail = json.dumps(out)
    try:
        d = ROOT / "gpurun_out"
        d.mkdir(exist_ok=True)
        (d / "bench_detail.json").write_text(detail + "\n")
    except OSError as e:  # (a read-only tree must not cost the line)
        print(f"bench.py: could not write gpurun_out/bench_detail.json: {e}", file=sys.stderr)
    print("bench detail: " + detail, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def _traffic_profile():
    """the committed rocprofv3 FETCH_SIZE / WRITE_SIZE summary of this same command (made by
    tools/make_profile_summary.py from separate --pmc passes): newest round first"""
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json"):
        f = ROOT / "profiles" / name
        if f.exists():
            return name, json.loads(f.read_text())
    return None, None


def measured_traffic(kernel, H, W, S):
    """HBM bytes per launch of `kernel` from the committed profile (NOT measured in this run: counters need their
    own rocprofv3 passes); None when the workload differs from the profiled one."""
    name, prof = _traffic_profile()
    if prof is None or (H, W, S) != (1080, 1920, 24):
        return None
    import re
    for k, v in prof["kernels"].items():
        if re.sub(r"<.*>", "", k).replace("_kernel", "") == kernel:
            return v["hbm_bytes_per_launch"]
    return None


def measured_view_traffic(H, W, S):
    """HBM bytes per view of all kernels together, from the same committed profile (None for other workloads)"""
    name, prof = _traffic_profile()
    if prof is None or (H, W, S) != (1080, 1920, 24):
        return None
    return prof.get("total_hbm_bytes_per_view")


def rocprof_avg_us(kernel):
    """average duration (us) of `kernel` in the committed rocprofv3 --kernel-trace --stats summaries: one view in flight on one
    stream (the mode of this file's HIP-event pass; tools/r06_stats_one_view.sh) and three lanes with second streams
    (tools/profile_round.sh).  rocprofv3 reports 4.6-5.0 us for ANY dispatch on this chip, an empty kernel included
    (tools/r06_latency_trace.sh), which the event pass's empty-bracket subtraction removes: the traced figure of a short kernel
    is its event figure + ~4.5 us.  {} when the files are missing."""
    import re
    out = {}
    for key, name in (("one_view_avg_us", "r06_kernel_stats_one_view.csv"), ("three_lanes_avg_us", "r06_kernel_stats.csv")):
        f = ROOT / "profiles" / name
        if not f.exists():
            continue
        best = None
        for ln in f.read_text().splitlines():
            if ln.startswith("#") or ln.startswith("kernel,"):
                continue
            parts = ln.rsplit(",", 4)  # (kernel names may contain commas: template arguments)
            if len(parts) == 5 and re.sub(r"<.*>", "", parts[0]).replace("_kernel", "") == kernel:
                if best is None or int(parts[1]) > best[0]:
                    best = (int(parts[1]), float(parts[3]))
        if best:
            out[key] = best[1]
            out[key.replace("_avg_us", "_source")] = f"profiles/{name}"
    return out


def pmc_instruction_profile():
    """per-kernel instruction counters per launch (SQ_INSTS_VALU / SALU / LDS wave-instructions, busy and wait
    cycles) from the committed rocprofv3 --pmc summary of this command, or {}"""
    for name in ("r06_pmc_instructions.json", "r05_pmc_instructions.json", "r04_pmc_instructions.json", "r03_pmc_instructions.json"):
        f = ROOT / "profiles" / name
        if f.exists():
            return name, json.loads(f.read_text())
    return None, {}


def valu_mix_profile():
    """opcode-class mix of the hot kernels' loop bodies and the measured issue cost per class (tools/valu_mix.py), or {}"""
    f = ROOT / "profiles" / "r05_valu_mix.json"
    return json.loads(f.read_text()) if f.exists() else {}


def gnt_pmc_profile():
    """matrix-pipe busy fraction per GNT kernel from the committed rocprofv3 --pmc summary (tools/pmc_gnt.sh; its header
    table: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x the kernel's duration)); (None, {}) when the file is missing"""
    import re
    for name in ("r06_gnt_bf16x3_pmc.txt", "r05_gnt_bf16x3_pmc.txt", "r04_gnt_bf16x3_pmc.txt"):
        f = ROOT / "profiles" / name
        if not f.exists():
            continue
        res = {}
        for ln in f.read_text().splitlines():
            m = re.match(r"^#\s+(gnt_\w+(?:<[^>]*>)?)\s+([0-9.]+)\s+\(\s*([0-9.]+)\)\s+([0-9.]+) M\s+([0-9.]+) M\s+([0-9.]+) %", ln)
            if m:
                res[m.group(1)] = {"mfma_busy_frac": round(float(m.group(6)) / 100.0, 4), "us_per_dispatch": float(m.group(2)),
                                   "mfma_insts_M": float(m.group(4)), "vector_insts_M": float(m.group(5))}
        return name, res
    return None, {}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed views per rank")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--views", type=int, default=4, help="distinct target views kept resident and cycled")
    ap.add_argument("--scene", choices=list(SCENES), default="nominal",
                    help="statistics of the synthetic video (pgdvs_amd.synth.make_video); the headline is 'nominal'")
    ap.add_argument("--no-scene-sweep", action="store_true", help="skip the `variants.scenes` objects (the other scenes at this size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event per-kernel pass")
    ap.add_argument("--pts-per-pixel", type=int, default=3)
    ap.add_argument("--no-outlier", action="store_true", help="dyn_pcl_remove_outlier=false (YAML default)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent target views rendered concurrently, each on its own HIP stream; 0 (default): one GPU "
                         "measures a few counts during warm-up, several ranks take the fixed default (no probe collectives)")
    ap.add_argument("--side-stream", action=argparse.BooleanOptionalAction, default=True,
                    help="give every lane a second stream for the dynamic branch's geometry (forked / joined inside the native "
                         "call); --no-side-stream: one stream per lane")
    ap.add_argument("--place-streams", action=argparse.BooleanOptionalAction, default=True,
                    help="pick the lanes' streams by hardware queue (runtime.stream_queue_groups: a probe of ~0.1 s with spin kernels "
                         "before anything is timed); --no-place-streams: creation order, as in rounds 1-4")
    ap.add_argument("--per-op", action="store_true",
                    help="rounds 1-3 arrangement: ~85 C-ABI calls per view enqueued from Python instead of ONE native call (A/B)")
    ap.add_argument("--run-ahead", type=int, default=6, help="views the host may have enqueued beyond the last finished one")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="seconds after which a rank that has not finished exits non-zero (a lost peer must not hang the job)")
    ap.add_argument("--gnt-rays", type=int, default=1024, help="rays of the GNT sub-benchmark chunk (0 = skip)")
    ap.add_argument("--latency-only", type=int, default=0,
                    help="diagnostic (tools/r06_latency_trace.sh): after the warm-up render N views one at a time with a second stream for the "
                         "dynamic branch and N on one stream, print the wall-clock medians to stderr and exit (no JSON line)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise the launcher, the view sharding, the per-step gather and the timing "
                         "protocol on CPU tensors over gloo (tests); prints a line with dry_run=true and value=null")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="(dry run) this rank exits non-zero: launcher error path")
    ap.add_argument("--dry-hang-rank", type=int, default=-1, help="(dry run) this rank never reaches the timed loop: watchdog path")
    return ap.parse_args()


# Several ranks: no arrangement probe (it would need collectives to agree on its outcome), one fixed arrangement per rank:
# four lanes, one stream each, every stream on a hardware queue of its own as far as the rank's LOCAL queue probe
# (runtime.stream_queue_groups: spin kernels on this rank's GPU, no collective in it) can tell them apart; creation order when the
# probe fails, and the line says so (config.stream_placement_per_rank)
DEFAULT_LANES_MULTI_RANK = 4
DEFAULT_ARRANGEMENT_MULTI_RANK = (4, False, True)  # (lanes, second stream per lane, streams by hardware queue)


def ring_slots_for(run_ahead, n_lanes):
    """receive-ring slots per rank on rank 0 (and local slots on every rank): the views in flight, the host's run-ahead and a spare"""
    return max(run_ahead, n_lanes + 1) + n_lanes + 2
SCENES = ("nominal", "wide_baseline", "noisy_depth")


def start_watchdog(seconds, rank):
    """A rank that is still alive after `seconds` exits with status 124 from a daemon thread -- a peer that died or
    never arrived leaves the others inside a collective that no Python exception can leave (reference:
    trainer_pgdvs.py relies on the launcher for this).  Fresh-process semantics: the process ENDS (os._exit), nothing is
    re-executed; torch.distributed.run then tears the other ranks down and the launcher returns non-zero."""
    import threading

    if seconds <= 0:
        return None

    def fire():
        print(f"bench.py: rank {rank} did not finish within {seconds:.0f} s (--rank-timeout): giving up", file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher around it: start N fresh rank processes (one per
    GPU) through torch.distributed.run, as the reference's run.py:158-176 spawns its own workers.  This
    process has made no GPU call (importing torch and parsing flags do not initialise HIP) and makes none:
    it only waits, relays the children's output (rank 0 prints the JSON line) and returns their exit code,
    which is non-zero if any rank failed."""
    import subprocess

    n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if not args.dry_run and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL fails on this host driver without it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_main(args, world, rank):
    """The N-rank protocol of the benchmark without the renderer: same sharding (`(j + rank) % n_views`),
    same per-step asynchronous gather into a preallocated stack, same barrier / max-over-ranks timing,
    same JSON keys -- on CPU tensors over gloo.  Test infrastructure for the launcher; measures nothing."""
    from pgdvs_amd import dist as pdist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus, f"world size {dist.get_world_size()} != --gpus {args.gpus}"
    if rank == args.dry_fail_rank:
        print(f"rank {rank}: failing on request", file=sys.stderr)
        sys.exit(3)
    if rank == args.dry_hang_rank:  # a rank that never joins its peers again: only the watchdogs end the job
        time.sleep(10 ** 6)
    like = torch.empty(1, 3, 4, 6)
    start_watchdog(args.rank_timeout, rank)
    # (the ring of the real run with the fixed lane count several ranks use: main() sizes it with the same function)
    ring = min(args.steps, ring_slots_for(args.run_ahead, DEFAULT_LANES_MULTI_RANK if world > 1 else max(1, args.inflight)))
    gather = pdist.AsyncImageGather(dst=0, n_steps=args.steps, like=like, ring=ring)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for j in range(args.steps):
        gather.submit(torch.full_like(like, float(j * world + rank)))  # image of view j * world + rank
    out = gather.finish()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
    if rank == 0:
        # every view arrived exactly once: checksum of view v = 72 v (72 pixels of value v)
        assert out["sums"].flatten().tolist() == [72.0 * v for v in range(args.steps * world)], out["sums"]
        assert [float(x) for x in out["tail"][:, 0, 0, 0]] == [float(v) for v in range(out["tail_first_step"] * world, args.steps * world)]
        print(json.dumps({"metric": "novel-view frames/s at 1080p x 24 src frames; achieved HBM GB/s vs gfx950 peak",
                          "value": None, "unit": "frames/s", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(max(per_rank) / max(args.steps, 1) * 1e3, 3),
                          "scaling": "weak", "views_gathered": int(out["sums"].numel()), "receive_ring_slots": ring,
                          "per_rank_seconds": [round(x, 4) for x in per_rank],
                          "gather_bytes_to_rank0": int(like.numel() * 4 * args.steps * (world - 1))}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def algorithmic_bytes(name, H, W, S, n_static, n_dyn, K):
    """Algorithmic HBM bytes of ONE launch of kernel `name` (kernel table):
    reference fp32 layouts, every input read once and every output written once."""
    P = H * W
    n0 = min(n_static, P)  # frame 0 appends (nearly) every static pixel; the later frames share the rest
    table = {
        # A12: new points read once (packed xyz, 12 B) and one occupancy byte stamped per later frame
        "agg_push0": n0 * (12 + (S - 1)) if S > 1 else 0,
        # one later frame: mask + own occupancy map read, selection bits written, depth of the new points, one byte stamped
        # per new point and later frame
        "agg_step": 2 * P + P / 8 + ((n_static - n0) / max(S - 1, 1)) * (4 + (S - 1) / 2.0) if S > 1 else 0,
        "agg_count": (S - 1) * P / 8,
        # rows of all later frames: selection bits; depth, rgb -> cloud row + xyz copy
        "agg_rows": (S - 1) * P / 8 + (n_static - n0) * (4 + 12 + 24 + 12),
        "agg_select": P + n0 * (4 + 12 + 24 + 12),  # frame 0: mask; depth, rgb -> cloud row + xyz copy
        "compact_count": P,
        "compact_scatter": P + 4 * P * 0.5,
        "raster_project_count": n_static * 12,       # xyz in, tile counters only
        "raster_fill": n_static * (12 + 16 * 2.8),   # xyz in, 16-byte list entries out (2.8 tiles per point at this radius)
        "raster_tile": n_static * 2.8 * 16 + P * 16 + P * K * 12,
        "dyn_warp": P * (4 + 1 + 1) + n_dyn * (8 + 4 + 12 + 4 + 4 * 12 + 24),
        "project_flow_dense": P * (1 + 12) + n_dyn * 12,
        "dyn_splat_scatter": P * (12 + 8 + 8 + 4 + 12 + 4 * 12) + P * 4 * 4 + n_dyn * 4 * 4 * 5,
        "dyn_splat_finish": P * (20 + 12 + 16 + 36),
        "knn_mean_dist": n_dyn * 12 + n_dyn * 4,
        "grid_query": n_dyn * 16 + n_dyn * 4,
        "grid_query_tpq": n_dyn * 16 + n_dyn * 4,  # cell-sorted points in, one mean per point out
        "grid_fallback": n_dyn * 16,
        "grid_count": n_dyn * 16, "grid_fill": n_dyn * 32, "stat_pass": n_dyn * 4,
        "gather_rows": n_dyn * (4 + 12 + 12),
        "scatter_keep": n_dyn * 6,
    }
    return float(table.get(name, 0.0))


def gnt_full_frame(dev, H=288, W=550, V=10, chunk=1024):
    """seconds per target view of PGDVSRenderer.forward with static_renderer=gnt (random-init
    8-layer GNT + ResUNet features + dynamic splat + composite) on synthetic inputs"""
    from pgdvs_amd import synth
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer

    cfg = load_config(static_renderer="gnt")
    rc = cfg.engine.engine_cfg.render_cfg
    rc.chunk_size, rc.n_coarse_samples_per_ray = chunk, 256
    rc.gnt_use_masked_spatial_src, rc.gnt_use_dyn_mask = False, True
    torch.manual_seed(0)
    model = PGDVSRenderer(cfg, render_cfg=rc).to(dev).eval()
    video = synth.make_video(V, H, W, seed=3)
    d = synth.to_torch(synth.make_view(video, V // 2, seed=1), dev)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)  # noqa: E731
    d["rgb_src_spatial"] = T(video["rgbs"])[None]
    d["dyn_mask_src_spatial"] = T(video["dyn_masks"].astype(np.float32))[None, ..., None]
    d["flat_cam_src_spatial"] = T(np.stack([synth.flat_cam(H, W, video["K3s"][i], video["c2ws"][i]) for i in range(V)]))[None]
    d["depth_range"] = T(np.array([[0.8, 5.0]]))
    with torch.no_grad():
        ret = model.forward(d, render_cfg=rc, disable_tqdm=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(ret["combined_rgb"]).all())
        # parity, not only finiteness: the same view at render_stride 32 (rays at integer pixel centres: a subset of the full
        # frame's, 9 x 18 of them) through the torch statement of the network (fused kernels off) against the full frame's
        # pixels; 2e-4: the ResUNet's MIOpen convolutions are not bit-reproducible from call to call
        import copy

        from pgdvs_amd import ops as _ops

        rc_s = copy.deepcopy(rc)
        rc_s.render_stride = 32
        _ops._GNT_VIEW_ENABLED = False
        try:
            mir = model.forward(d, render_cfg=rc_s, disable_tqdm=True)["static_coarse_rgb"]
        finally:
            _ops._GNT_VIEW_ENABLED = True
        sub = ret["static_coarse_rgb"][:, :, ::32, ::32]
        assert sub.shape == mir.shape, (sub.shape, mir.shape)
        parity = float((sub - mir).abs().max())
        assert parity <= 2e-4, f"GNT full frame: kernels vs torch statement on {mir[0, 0].numel()} rays: max|d| = {parity}"
        t0 = time.perf_counter()
        model.forward(d, render_cfg=rc, disable_tqdm=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # where the time goes: the same forward once more with an event pair per stage (BaseRenderer.stage_events)
        model.static_renderer.stage_events = ev = {}
        t1 = time.perf_counter()
        model.forward(d, render_cfg=rc, disable_tqdm=True)
        torch.cuda.synchronize()
        dt_ev = time.perf_counter() - t1
        model.static_renderer.stage_events = None
    ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in ev.items()}
    n_chunks = len(ev.get("transformer", []))
    breakdown = {
        "chunks": n_chunks, "resunet_ms": round(ms.get("features", 0.0), 2),
        "gather_A13_ms_sum": round(ms.get("gather", 0.0), 2),
        "transformer_A14_and_ray_reductions_ms_sum": round(ms.get("transformer", 0.0), 2),
        "rest_ms": round(dt_ev * 1e3 - sum(ms.get(k, 0.0) for k in ("features", "gather", "transformer")), 2),
        "seconds_with_events": round(dt_ev, 3),
        "note": "event pairs around the stages of one forward (ResUNet on the source views; per chunk the epipolar gather and "
                "GNT.forward with the per-ray reductions); rest = ray set-up, dynamic branch, composite, output concatenation, "
                "gaps between launches.  Enqueueing the gather of chunk i+1 on a side stream ahead of the transformer of chunk i "
                "was measured (round 4) and gains nothing: A14's kernels are persistent workgroups that own every CU's "
                "registers, the gather only runs in the seams either way"}
    return {"seconds_per_view": round(dt, 3), "frames_per_s": round(1.0 / dt, 3), "height": H, "width": W, "spatial_views": V,
            "temporal_views": 2, "samples_per_ray": 256, "chunk_rays": chunk, "weights": "random init", "breakdown": breakdown,
            "parity_vs_torch_statement": {"rays": int(mir[0, 0].numel()), "max_abs_diff_static_rgb": parity, "bound": 2e-4},
            "reference_context": "the reference states ~2 days on 8 A100 for 15 840 such views incl. data loading and metrics "
                                 "(docs/BENCHMARK_NVIDIA.md:148-149): ~87 s per view per GPU; not the same hardware or scope"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # before anything touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU"
    if args.dry_run:
        return dry_main(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    start_watchdog(args.rank_timeout, rank)
    # test hook (tests/test_gpu_round2.py): every rank on GPU 0 with gloo moving the device tensors, so that the
    # N-rank code of this file (per-step gather, max-over-ranks timing) runs on a 1-GPU box
    shared_gpu_test = os.environ.get("PGDVS_BENCH_SHARED_GPU_TEST") == "1"
    if shared_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus, f"world size {dist.get_world_size()} != --gpus {args.gpus}"

    from pgdvs_amd import _lib, dist as pdist, harness, ops, synth
    from pgdvs_amd.instantiate import load_config
    from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer
    from pgdvs_amd.runtime import ResidentVideoRenderer

    lib = _lib.load()
    H, W, S, K = args.height, args.width, args.frames, args.pts_per_pixel
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    cfg = load_config(static_renderer="geo", overrides={
        "engine.engine_cfg.render_cfg.dyn_pcl_remove_outlier": not args.no_outlier,
        "engine.engine_cfg.render_cfg.st_render_pcl_pts_per_pixel": K,
    })
    rc = cfg.engine.engine_cfg.render_cfg
    model = PGDVSRenderer(cfg, render_cfg=rc, softsplat_metric_abs_alpha=100.0).to(dev).eval()
    n_views = max(1, min(args.views, S - 1))
    view_ids = [int(round(j * (S - 2) / max(n_views - 1, 1))) for j in range(n_views)]

    def load_scene(scene, size=None):
        """synthetic, seeded inputs (no datasets offline) -> resident in HBM; the renderer around them
        (size: (H, W, S) of another BASELINE configuration; default: this run's)"""
        H_, W_, S_ = size or (H, W, S)
        video = synth.make_video(S_, H_, W_, seed=1234, scene=scene)
        rgbs, depths, masks = T(video["rgbs"]), T(video["depths"]), T(video["dyn_masks"]).view(torch.uint8)
        nv_ = max(1, min(args.views, S_ - 1))
        ids_ = [int(round(j * (S_ - 2) / max(nv_ - 1, 1))) for j in range(nv_)]
        views = [synth.to_torch(synth.make_view(video, i, frac=0.4, seed=5), dev) for i in ids_]
        # The timed views carry NO injected noise: like the reference (torch.randn_like per forward,
        # pgdvs_renderer_dyn.py:177-182) every step draws its own -- in the splat kernel, where it is consumed.  One
        # view keeps the synthetic generator's field for the checks that need two renders to agree.
        check_view = dict(views[0])
        for d_ in views:
            d_.pop("static_noise", None)
        rvr = ResidentVideoRenderer(model, rc, rgbs, depths, masks, video["K3s"], video["c2ws"], lanes=1,
                                    side_streams=args.side_stream, native=not args.per_op)
        return video, views, check_view, rvr

    video, views, check_view, rvr = load_scene(args.scene)
    K3s, c2ws = video["K3s"], video["c2ws"]
    cap = S * H * W

    # Views are independent (the reference shards them over ranks): `inflight` of them are kept in flight per GPU, each
    # on its own stream, so one view's launch-bound chains fill the gaps of another's.  Every view still runs the
    # complete path.  One GPU: the count is measured during warm-up; several ranks: a fixed default, because a probe
    # needs collectives to agree on its outcome and the N > 1 path must not depend on anything that has never run with
    # peers (first contact with an 8-GPU node happens in the driver's run).
    # (lanes, second stream per lane, streams picked by hardware queue): four hardware queues run side by side on this chip
    # (tools/overlap_probe.py) and two streams that share one run one behind the other, so the arrangement matters as
    # much as the count -- three lanes with second streams in creation order (round 4's), four single-stream lanes on four
    # distinct queues, two lanes whose four streams have a queue each (round 6, measured and left out: three lanes on three
    # queues with their second streams sharing the fourth -- 0.87-0.89 ms per view against 0.79 for the four single-stream lanes)
    lane_candidates = [(3, True, False), (4, False, True), (2, True, True)] if args.place_streams else [(2, True, False), (3, True, False), (5, True, False)]
    if not args.side_stream:
        lane_candidates = [(k, False, p) for k, _, p in lane_candidates]
    auto_lanes = args.inflight <= 0 and world == 1
    if args.inflight <= 0 and world > 1:
        args.inflight = DEFAULT_LANES_MULTI_RANK
    if auto_lanes:
        lane_cfg = max(lane_candidates, key=lambda c: c[0])
    elif world > 1 and args.inflight == DEFAULT_LANES_MULTI_RANK and args.place_streams:
        lane_cfg = DEFAULT_ARRANGEMENT_MULTI_RANK
    else:
        lane_cfg = (max(1, args.inflight), args.side_stream, args.place_streams and not args.side_stream)
    n_lanes = lane_cfg[0]
    rvr.set_lanes(*lane_cfg)
    base_run_ahead = args.run_ahead
    args.run_ahead = max(base_run_ahead, n_lanes + 1)  # the bound must leave every lane a view to work on

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    # the first view sizes the cloud buffers and the rasteriser's workspace for the views that follow (one host read of
    # the count, untimed)
    n_first = rvr.calibrate(views[rank % n_views])
    if os.environ.get("PGDVS_BENCH_NO_ROW_BOUND"):  # diagnostic: capacity-sized buffers and workspaces as in round 2
        rvr.row_bound = None
    torch.cuda.synchronize()

    host_enqueue = [0.0]
    host_wait = [0.0]  # part of host_enqueue spent blocked on the run-ahead bound (the GPU is behind)
    host_native = [0.0]  # seconds inside pgdvs_view_geo_forward (the C ABI's own clock)
    mem_probe = {}
    gather_boxes = {}  # image shape -> AsyncImageGather
    ctl_stream = torch.cuda.Stream(device=dev)
    # ring slots: the views in flight at the deepest lane count tried, the host's run-ahead and a spare
    ring_slots = ring_slots_for(base_run_ahead, n_lanes)
    last = {}

    def timed(n_steps, profile=False, rv=None, vs=None):
        """the benchmark loop: n_steps views through `rv` (default: the headline scene's renderer), image of step j
        gathered to rank 0 while step j+1 renders, bounded host run-ahead; returns wall seconds"""
        rv = rv or rvr
        vs = vs or views
        n_vs = len(vs)
        lib.pgdvs_prof_enable(1 if profile else 0)
        # step j's image travels while step j+1 renders; rank 0 receives into one ring allocated before the first loop
        # and reused by every later one (bounded memory; rank 0 checksums every view as its slot comes up for reuse)
        like = ref_img if rv is rvr or (rv.H, rv.W) == (H, W) else torch.empty(1, 3, rv.H, rv.W, device=dev)
        key = tuple(like.shape)
        if key not in gather_boxes or n_steps > gather_boxes[key].capacity_steps:
            gather_boxes[key] = pdist.AsyncImageGather(dst=0, n_steps=max(n_steps, 256), like=like, ring=ring_slots)
        gather = gather_boxes[key].reset(n_steps)
        barrier()
        torch.cuda.synchronize()
        mem_probe["before"] = torch.cuda.memory_stats(dev)
        ops.view_geo_host_stats()
        timeline = os.environ.get("PGDVS_BENCH_TIMELINE") == "1"  # diagnostic: when every view of the loop finished
        if timeline:
            ev_t0 = torch.cuda.Event(enable_timing=True)
            ev_t0.record(ctl_stream)
            enq_t = []
        t0 = time.perf_counter()
        done = []
        host_wait[0] = 0.0
        # (everything the loop itself enqueues -- slot retirement, joins, the final checksums -- runs on a control stream
        # of its own, never on the null stream)
        with torch.cuda.stream(ctl_stream):
            for j in range(n_steps):
                # bounded run-ahead: an unbounded loop gets tens of views ahead, and every view enqueued but not yet
                # executed pins its output blocks: the allocator's pools then grow by hipMalloc calls in the middle of the
                # timed region, each of which drains the pipeline
                if len(done) >= args.run_ahead:
                    w0 = time.perf_counter()
                    done[j - args.run_ahead].synchronize()
                    host_wait[0] += time.perf_counter() - w0
                # per-kernel HIP events need one view at a time, so that a kernel's duration is its own and not the
                # queueing behind the other lanes' kernels
                ret, main = rv.render(vs[(j + rank) % n_vs], 0 if profile else j, out=gather.slot(), use_side=not profile)
                with torch.cuda.stream(main):
                    gather.submit(ret["combined_rgb"])
                    ev = torch.cuda.Event(enable_timing=timeline)
                    ev.record()
                done.append(ev)
                if timeline:
                    enq_t.append(time.perf_counter() - t0)
            host_enqueue[0] = time.perf_counter() - t0  # host time to enqueue everything (incl. the waits of the run-ahead bound)
            rv.join()
            gathered = gather.finish(tail=False)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        if timeline:
            print(f"timeline ({n_steps} views, {rv.n_lanes} lanes): wall {1e3 * (t1 - t0):.3f} ms; view: host enqueued at / GPU finished at (ms): "
                  + " ".join(f"{1e3 * a:.2f}/{ev_t0.elapsed_time(b):.2f}" for a, b in zip(enq_t, done)), file=sys.stderr)
        host_native[0] = ops.view_geo_host_stats()[1]
        mem_probe["after"] = torch.cuda.memory_stats(dev)
        lib.pgdvs_prof_enable(0)
        last["ret"], last["gathered"] = ret, gathered
        return t1 - t0

    if os.environ.get("PGDVS_BENCH_HOST_PROFILE"):  # diagnostic: where the host time of a view goes
        import cProfile
        import pstats

        for j in range(6):
            rvr.render(views[j % n_views], 0)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for j in range(60):
            rvr.render(views[j % n_views], 0)
            if j % 6 == 5:
                torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)

    ref_img = None
    for j in range(max(args.warmup, n_lanes)):
        ret, _ = rvr.render(views[(j + rank) % n_views], j)
        ref_img = ret["combined_rgb"] if j == 0 else ref_img
    rvr.join()
    torch.cuda.synchronize()
    if world > 1:
        # RCCL sets its point-to-point connections up at the first send / receive between two ranks:
        # have that (and the receive path on rank 0) behind us before the timed region
        g = pdist.AsyncImageGather(dst=0, n_steps=2, like=ref_img)
        g.submit(ref_img)
        g.submit(ref_img)
        g.finish()
        del g
        torch.cuda.synchronize()
        barrier()
    if args.latency_only > 0:
        for use_side in (True, False):
            lat_ = []
            for j in range(args.latency_only):
                torch.cuda.synchronize()
                l0 = time.perf_counter()
                rvr.render(views[j % n_views], 0, use_side=use_side)
                rvr.join()
                torch.cuda.synchronize()
                lat_.append((time.perf_counter() - l0) * 1e3)
            lat_.sort()
            print(f"latency-only: second stream {use_side}: median {lat_[len(lat_) // 2]:.3f} ms, min {lat_[0]:.3f} ms over {len(lat_)} views", file=sys.stderr)
        return
    lanes_note = f"{n_lanes} (" + ("--inflight" if world == 1 or args.inflight != DEFAULT_LANES_MULTI_RANK else
                                   "fixed arrangement for several ranks, streams placed by each rank's own queue probe: no probe collectives") + ")"
    if auto_lanes:
        # How many views in flight, on which streams?  Measured, not guessed: the candidate arrangements are timed through the
        # same loop as the headline, each on as many views as the timed region will render (filling and draining k lanes is
        # part of a short run)
        trial = {}
        for cfg_ in lane_candidates:  # (every arrangement's streams, workspaces and allocator pools exist before any is timed)
            rvr.set_lanes(*cfg_)
            args.run_ahead = max(base_run_ahead, cfg_[0] + 1)
            timed(2 * cfg_[0])
        for cfg_ in lane_candidates:
            k = cfg_[0]
            rvr.set_lanes(*cfg_)
            args.run_ahead = max(base_run_ahead, k + 1)
            timed(12)  # (rehearsal right in front of the measurement: the chip's clocks after the switch)
            n_probe = max(2 * k, min(8 * k, args.steps)) if args.steps > 32 else max(2 * k, args.steps)
            trial[cfg_] = timed(n_probe) / n_probe
        # (arrangements within 1.5 % of the fastest: the one with the fewest views in flight -- the shortest fill and drain)
        best_t = min(trial.values())
        lane_cfg = min((c for c in lane_candidates if trial[c] <= 1.015 * best_t), key=lambda c: c[0])
        n_lanes = lane_cfg[0]
        rvr.set_lanes(*lane_cfg)
        rvr.drop_other_lane_sets()  # (the probe's other arrangements held ~2 GB of workspace per lane)
        torch.cuda.reset_peak_memory_stats(dev)  # (`memory.reserved_GB_peak` is the arrangement's that runs, not the probe's)
        args.run_ahead = max(base_run_ahead, n_lanes + 1)
        name_ = lambda c: f"{c[0]} lanes{' + second streams' if c[1] else ''}{', streams by hardware queue' if c[2] else ''}"  # noqa: E731
        lanes_note = ("auto (probed on min(8 k, --steps) views each): " + "; ".join(f"{name_(c)} {trial[c] * 1e3:.3f} ms/view" for c in lane_candidates)
                      + f" -> {name_(lane_cfg)}")

    import gc

    # untimed rehearsal through the same loop: the allocator's pools reach the state the bounded
    # run-ahead needs, so that the timed region allocates from them only (`device_mallocs_in_timed_region`)
    # The collector runs BEFORE the rehearsal, not between it and the timed region: a collection walks the whole heap
    # (~0.1 s of host time with the GPU idle, the host's caches flushed), and the 20-view timed region that followed it
    # ran its first view's enqueue in 0.98 ms instead of 0.48 and every view ~6 % slower than the same loop a moment
    # later (the chip's clocks after an idle gap; `PGDVS_BENCH_TIMELINE=1` prints when every view was enqueued / finished)
    gc.collect()
    gc.disable()  # no collector pauses inside the timed loop
    timed(max(2 * args.run_ahead + n_lanes, 24))
    elapsed = timed(args.steps)
    gc.enable()
    gathered, cnt = last["gathered"], last["ret"]["st_pcl_rgb_count"]
    raster_status = last["ret"].get("geo_static_raster_status", None)
    ms0, ms1 = mem_probe["before"], mem_probe["after"]
    host_ms = host_enqueue[0] / args.steps * 1e3  # (of the headline loop: the steady-state loop below overwrites the counters)
    host_wait_ms = host_wait[0] / args.steps * 1e3
    host_native_ms = host_native[0] / args.steps * 1e3
    # a short timed region (the driver's --steps 20) spends a visible share filling and draining the lanes: the
    # steady-state rate of the same loop is reported beside it (never `value`)
    steady = None
    if args.steps < 100 and world == 1:
        e2 = timed(200)
        steady = {"frames_per_s": round(200 / e2, 2), "steps": 200,
                  "note": "same loop and lane count over 200 views, measured right after the timed region; not the headline value"}
    mem_note = {"device_mallocs_in_timed_region": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                "reserved_GB_peak": round(ms1.get("reserved_bytes.all.peak", 0) / 1e9, 2),
                "allocated_GB_peak": round(ms1.get("allocated_bytes.all.peak", 0) / 1e9, 2)}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank_s = [elapsed]
    if world > 1:
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_s = [float(x.item()) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # who took part (first contact with an 8-GPU node: a silent rank / device mismatch must show in the line)
    placement = ("by hardware queue " + str(rvr.queue_groups)) if (rvr.place_streams and rvr.queue_groups and len(rvr.queue_groups) > 1) else \
        ("creation order (queue probe failed)" if rvr.place_streams else "creation order")
    peers = [(rank, local_rank, torch.cuda.get_device_name(dev), placement)]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, peers[0])
        peers = sorted(got)
        assert [p_[0] for p_ in peers] == list(range(world)), f"ranks seen: {peers}"
    if rank == 0 and isinstance(gathered, dict):
        sums = gathered["sums"]
        assert tuple(sums.shape) == (args.steps, world) and bool(torch.isfinite(sums).all()) and bool((sums != 0).all()), \
            "a gathered view is missing or empty"
    n_static = ops.checked_count(cnt, "pgdvs_static_aggregate")
    ops.check_raster_status(raster_status)  # (the last view's status word; every view renders the same cloud)
    assert rvr.row_bound is None or n_static < rvr.row_bound, \
        f"the static cloud ({n_static} rows) filled its buffer of {rvr.row_bound} rows: rows may have been dropped"
    # concurrency must not change results: the same view (fixed noise field) alone on one lane and on every lane at once
    torch.cuda.synchronize()
    alone = rvr.render(check_view, 0)[0]["combined_rgb"]
    rvr.join()
    torch.cuda.synchronize()
    alone = alone.clone()  # (after the join: this stream has not seen the lane's work before)
    together = [rvr.render(check_view, li)[0]["combined_rgb"] for li in range(n_lanes)]
    rvr.join()
    torch.cuda.synchronize()
    # (the splat accumulates with float atomics, so the comparison is to rounding, not bit-exact)
    assert all(torch.allclose(t_, alone, rtol=0, atol=1e-5) for t_ in together), "in-flight views disagree with the sequential result"
    del together
    n_dyn = int(views[0]["dyn_mask_src_temporal"][0, 0].sum().item())

    # ---------------- per-kernel durations with HIP events on the launch stream
    kernels = {}
    roofline = None
    roofline_kernels = None
    if not args.no_kernel_timing:
        n_prof = min(args.steps, 5)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.pgdvs_prof_report(buf, len(buf))  # clear
        timed(n_prof, profile=True)  # every rank takes part (barriers inside); rank 0 reports
        lib.pgdvs_prof_report(buf, len(buf))
        if rank == 0:
            for line in buf.value.decode().strip().splitlines():
                name, calls, total_ms = line.split()
                calls, total_ms = int(calls), float(total_ms)
                ab = algorithmic_bytes(name, H, W, S, n_static, n_dyn, K)
                avg_ms = total_ms / calls
                kernels[name] = {
                    "launches_per_step": calls / n_prof, "avg_ms": round(avg_ms, 5),
                    "ms_per_step": round(total_ms / n_prof, 4),
                    "alg_GBps": round(ab / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 and ab > 0 else None}
            # dominant kernel = the entry with the most time per view, CHAINS INCLUDED (the aggregation's 23 links are one
            # entry: 23 launches of ~11 us); every entry within 20 % of it is listed in roofline_kernels with its own
            # algorithmic bytes, launch time, fraction and counter traffic -- no eligibility rule
            top = max(v["ms_per_step"] for v in kernels.values())
            near = sorted((k for k, v in kernels.items() if v["ms_per_step"] >= 0.8 * top), key=lambda k: -kernels[k]["ms_per_step"])
            for k in ("agg_step", "grid_query_tpq", "raster_tile"):  # (the three the reviews follow, whatever their order on this box)
                if k in kernels and k not in near:
                    near.append(k)
            dom = near[0]
            pmc_name, pmc_all = pmc_instruction_profile()
            pmc = pmc_all.get("kernels", {}) if (H, W, S) == (1080, 1920, 24) else {}
            mix_all = valu_mix_profile()
            mix = mix_all.get("kernels", {})
            tprof_name, _ = _traffic_profile()

            def kernel_roofline(k):
                ab_ = algorithmic_bytes(k, H, W, S, n_static, n_dyn, K)
                ach_ = ab_ / (kernels[k]["avg_ms"] * 1e-3) / 1e9
                o = {"bound": "hbm", "alg_bytes": ab_, "avg_ms": kernels[k]["avg_ms"], "launches_per_view": kernels[k]["launches_per_step"],
                     "ms_per_view": kernels[k]["ms_per_step"], "achieved": round(ach_, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach_ / HBM_PEAK_GBS, 5), "traffic": measured_traffic(k, H, W, S)}
                # the same kernel in the committed rocprofv3 traces (dispatch overhead included; beside other lanes' kernels)
                tr_ = rocprof_avg_us("agg_push" if k == "agg_push0" else k) if (H, W, S) == (1080, 1920, 24) else {}
                if tr_:
                    o["rocprofv3"] = dict(tr_, **{kk.replace("_avg_us", "_frac"): round(ab_ / (v_ * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                                                  for kk, v_ in tr_.items() if kk.endswith("_avg_us")},
                                          note="avg_ms above is execution time (HIP events, the empty-launch bracket subtracted); rocprofv3 "
                                               "adds ~4.5 us to every dispatch, and beside two other lanes a kernel waits for memory longer")
                c = pmc.get(k)
                if c and c.get("SQ_INSTS_VALU"):
                    # vector-instruction issue on two bases: the nominal 2 cycles per wave64 instruction on a SIMD-32
                    # (MI355X_MICROARCH.md), and what THIS chip was measured to take per instruction of each opcode class
                    # (profiles/r03_valu_rate.txt) weighted by the class mix of the kernel's loop bodies (tools/valu_mix.py)
                    dur = kernels[k]["avg_ms"] * 1e-3
                    t_nom = c["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9)
                    o["valu_issue"] = {"valu_wave_insts_per_launch": c["SQ_INSTS_VALU"], "salu_wave_insts_per_launch": c.get("SQ_INSTS_SALU"),
                                       "valu_issue_frac": round(t_nom / dur, 4), "basis": "2 cycles per wave64 instruction at 2.4 GHz, 1024 SIMDs",
                                       "source": f"profiles/{pmc_name} (committed rocprofv3 --pmc pass of this command), duration from this run"}
                    m_ = mix.get(k)
                    if m_:
                        t_meas = c["SQ_INSTS_VALU"] * m_["ns_per_wave_instruction"] * 1e-9 / 1024
                        o["valu_issue"].update({"valu_issue_frac_measured_costs": round(t_meas / dur, 4),
                                                "ns_per_wave_instruction_measured": m_["ns_per_wave_instruction"], "opcode_mix_assumed": m_["mix"],
                                                "measured_basis": "profiles/r03_valu_rate.txt (ns per wave64 instruction per SIMD by opcode class, 8 "
                                                                  "wavefronts per SIMD) x the static class mix of the kernel's loop bodies "
                                                                  "(profiles/r05_valu_mix.json, tools/valu_mix.py)"})
                return o

            roofline_kernels = {k: kernel_roofline(k) for k in near}
            kind = ("a chain of short launches, one per source frame, each bound by its dependent global round trips (mask + map -> "
                    "selected pixels -> depth -> stamps), not by bandwidth" if kernels[dom]["launches_per_step"] > 4 else
                    "a VALU-bound search kernel, not an HBM stream (valu_issue)" if dom in ("raster_tile", "grid_query", "grid_query_tpq", "agg_push0")
                    else "see DESIGN.md section 4")
            roofline = dict(roofline_kernels[dom])
            roofline.update({"kernel": dom,
                             "traffic_source": f"profiles/{tprof_name} (committed rocprofv3 --pmc passes of this command; not this run)" if tprof_name else None,
                             "alg_bytes_per_launch": roofline_kernels[dom]["alg_bytes"], "avg_launch_ms": kernels[dom]["avg_ms"],
                             "event_bracket_overhead_ms": round(float(lib.pgdvs_prof_overhead_ms()), 5),
                             "co_dominant": {k: kernels[k]["ms_per_step"] for k in near},
                             "note": ("the entry with the most time per view (HIP events on the launch stream, one view at a time, the "
                                      "empty-launch bracket cost subtracted), chains of launches included: " + kind
                                      + "; the other entries within 20 % of it and the three the reviews follow are in roofline_kernels; "
                                        "the whole path's figure is roofline_path (DESIGN.md section 4)")})

    # ---------------- latency of ONE view (nothing else in flight): the throughput above comes from overlapping
    # `inflight` independent views; this is the time a single view takes from first launch to last kernel
    torch.cuda.synchronize()
    # (one lane with a second stream for the dynamic branch, whatever arrangement the throughput loop runs: a view alone is
    # the static branch's chain with the dynamic branch beside it)
    rvr.set_lanes(1, True, False)
    for j in range(3):
        rvr.render(views[j % n_views], 0)
    rvr.join()
    torch.cuda.synchronize()
    lat = []
    for j in range(12):  # (every rank, so that all ranks reach the end of the run together; rank 0 reports its own)
        l0 = time.perf_counter()
        rvr.render(views[j % n_views], 0)
        rvr.join()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - l0) * 1e3)
    lat = sorted(lat[2:])
    latency_ms = {"median": round(lat[len(lat) // 2], 3), "min": round(lat[0], 3), "views": len(lat),
                  "note": "one view in flight on one lane (second stream for the dynamic branch), wall clock around enqueue + synchronize"}
    rvr.set_lanes(*lane_cfg)

    # ---------------- what a drop-in caller of pgdvs.engines gets: the reference's evaluator renders ONE view, then
    # synchronises for its metrics (evaluator_pgdvs.py:36-188) -- the same workload through harness.eval_step, one view
    # at a time, ground truth = a source frame (the values do not matter for the timing)
    eval_loop = None
    if rank == 0 and world == 1:
        ev_views = []
        for d_ in views:
            e_ = dict(d_)
            e_["_st_pcl_video"] = dict(rvr.video, capacity=rvr.row_bound or cap)
            if rvr.row_bound is not None:
                e_["st_pcl_rgb_row_bound"] = rvr.row_bound
            e_["rgb_tgt"] = d_["rgb_src_temporal"][:, 0]
            e_["eval_mask"] = d_["dyn_mask_src_temporal"][:, 0].expand(-1, -1, -1, 3).contiguous()
            ev_views.append(e_)
        for j in range(3):
            harness.eval_step(model, ev_views[j % n_views], rc, device=dev)
        torch.cuda.synchronize()
        n_ev = max(8, min(args.steps, 40))
        e0 = time.perf_counter()
        for j in range(n_ev):
            md = harness.eval_step(model, ev_views[j % n_views], rc, device=dev)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - e0
        # where a step's host time goes (a second, instrumented pass: the timed one above carries no timers)
        harness.STAGE_SECONDS = st_ = {}
        for j in range(n_ev):
            harness.eval_step(model, ev_views[j % n_views], rc, device=dev)
        harness.STAGE_SECONDS = None
        # the renderer's plugin call alone on the same inputs, one view at a time (what eval_step adds is the rest)
        f0 = time.perf_counter()
        with torch.no_grad():
            for j in range(n_ev):
                model.forward(ev_views[j % n_views], render_cfg=rc, disable_tqdm=True, for_debug=False)
                torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - f0) / n_ev * 1e3
        eval_loop = {"frames_per_s": round(n_ev / e1, 2), "ms_per_view": round(e1 / n_ev * 1e3, 3), "views": n_ev,
                     "host_ms_per_view": {k_: round(v_ / n_ev * 1e3, 4) for k_, v_ in st_.items()},
                     "forward_and_synchronize_ms_per_view": round(fwd_ms, 3),
                     "psnr_full_last": round(float(md["eval/psnr_full_combined"]), 3),
                     "note": "pgdvs_amd.harness.eval_step per view (to-device, forward = one native call incl. A12, quantisation + "
                             "the three masked PSNRs in one pass, ONE host synchronisation, status words checked): the "
                             "reference evaluator's loop shape, one view in flight; `value` presumes a pipelined caller"}

    # ---------------- informational variants (never `value`)
    variants = None
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        variants = {}
        # what the reference's own flow would time per view -- it aggregates the static cloud ONCE per scene at dataset
        # construction (nvidia_eval_pure_geo.py:166-178) and only renders per target view
        cloud_c, cnt_c, xyz_c = ops.static_aggregate(rvr.video["rgbs"], rvr.video["depths"], rvr.video["dyn_masks"], K3s, c2ws,
                                                     capacity=rvr.row_bound or cap, return_xyz=True)

        def step_cached(j):
            main, _ = rvr.lanes[j % n_lanes]
            main.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(main), torch.no_grad():
                data = dict(views[j % n_views])
                data["st_pcl_rgb"], data["st_pcl_rgb_count"], data["st_pcl_xyz"] = cloud_c[None], cnt_c, xyz_c[None]
                if rvr.row_bound is not None:
                    data["st_pcl_rgb_row_bound"] = rvr.row_bound
                return model.forward(data, render_cfg=rc, disable_tqdm=True)["combined_rgb"]

        for j in range(2 * n_lanes):
            step_cached(j)
        rvr.join()
        torch.cuda.synchronize()
        v0 = time.perf_counter()
        nv = max(min(args.steps, 40), n_lanes)
        for j in range(nv):
            step_cached(j)
            if j % (2 * n_lanes) == 2 * n_lanes - 1:
                rvr.join()
                torch.cuda.synchronize()
        rvr.join()
        torch.cuda.synchronize()
        variants["static_cloud_aggregated_once_per_scene"] = {
            "frames_per_s": round(nv / (time.perf_counter() - v0), 2), "steps": nv,
            "note": "A12 outside the per-view loop, as the reference's dataset does; not the headline value"}
        del cloud_c, xyz_c

    # ---------------- scene statistics (never `value`): the same loop on videos whose statistics differ from the smooth
    # nominal scene -- a 12-camera rig cycled per frame (the NVIDIA monocular protocol: consecutive frames come from
    # different cameras, far more newly visible pixels per frame), noisy depth with flying pixels at object borders --
    # with the counters of every fast path's exit
    scene_stats = None
    if rank == 0 and world == 1 and not args.no_scene_sweep:
        scene_stats = {}

        def stats_of(rv, vs, label, arrangements=None):
            H, W, S = rv.H, rv.W, rv.S  # (another BASELINE configuration may be passed)
            rv.set_lanes(*lane_cfg)
            n0 = rv.calibrate(vs[0]) if rv is not rvr else n_static
            torch.cuda.synchronize()
            n_sc = max(min(args.steps, 40), 2 * n_lanes)
            cfg_used, tried = lane_cfg, None
            if arrangements:
                # another size, another best arrangement (at 540p four single-stream lanes beat two lanes with second
                # streams by a third): the headline's probe, repeated for this configuration
                tried = {}
                for c_ in arrangements:
                    rv.set_lanes(*c_)
                    timed(max(2 * args.run_ahead + c_[0], 12), rv=rv, vs=vs)
                    tried[c_] = timed(n_sc, rv=rv, vs=vs) / n_sc
                cfg_used = min(tried, key=tried.get)
            rv.set_lanes(*cfg_used)
            timed(2 * args.run_ahead + n_lanes, rv=rv, vs=vs)
            dt = timed(n_sc, rv=rv, vs=vs)
            # (two runs of n_sc views in this arrangement exist when the arrangements were probed -- the probe's and this one:
            # BOTH are reported, `frames_per_s` is this one's; a single run of 20 views at 1080p x 48 frames came out 35 % slow
            # once in ten lines, the allocator growing under it)
            ret_ = last["ret"]
            ops.check_raster_status(ret_.get("geo_static_raster_status", None))
            n_now = ops.checked_count(ret_["st_pcl_rgb_count"], "pgdvs_static_aggregate")
            assert rv.row_bound is None or n_now < rv.row_bound, f"{label}: the cloud filled its buffer"
            torch.cuda.synchronize()
            _, main1 = rv.render(vs[0], 0)
            rv.join()
            counters = model.view_counters(main1) if rv.native else None
            o = {"frames_per_s": round(n_sc / dt, 2), "ms_per_view": round(dt / n_sc * 1e3, 3), "steps": n_sc,
                 "static_points": n_now, "static_points_per_pixel": round(n_now / (H * W), 3),
                 "us_per_million_points": round(dt / n_sc * 1e6 / (n_now / 1e6), 1), "counters": counters}
            alg_ = (20 * S + 120) * H * W
            if tried:
                o["ms_per_view_runs"] = {"probe": round(tried[cfg_used] * 1e3, 3), "this": round(dt / n_sc * 1e3, 3)}
                o["arrangement"] = {"lanes": cfg_used[0], "second_streams": cfg_used[1], "streams_by_hardware_queue": cfg_used[2],
                                    "ms_per_view_tried": {f"{c_[0]} lanes{' + second streams' if c_[1] else ''}{', placed' if c_[2] else ''}": round(t_ * 1e3, 3)
                                                          for c_, t_ in tried.items()}}
            o["roofline_path"] = {"bound": "hbm", "alg_bytes_per_view": alg_, "achieved": round(alg_ * n_sc / dt / 1e9, 2),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_ * n_sc / dt / 1e9 / HBM_PEAK_GBS, 5)}
            return o

        scene_stats[args.scene] = stats_of(rvr, views, args.scene)
        for sc in SCENES:
            if sc == args.scene:
                continue
            try:
                _, vs_, _, rv_ = load_scene(sc)
                scene_stats[sc] = stats_of(rv_, vs_, sc)
                del vs_, rv_
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- the headline does not depend on it; the line says what failed
                scene_stats[sc] = {"error": f"{type(e).__name__}: {e}"}
        nom = scene_stats[args.scene]["us_per_million_points"]
        for sc, o in scene_stats.items():
            if "us_per_million_points" in o:
                o["per_point_cost_vs_headline_scene"] = round(o["us_per_million_points"] / nom, 3)
        if variants is None:
            variants = {}
        variants["scenes"] = scene_stats
        rvr.set_lanes(*lane_cfg)
        # ---------------- the other single-GPU BASELINE configurations through the SAME loop (never `value`):
        # configs[1] 960 x 540 x 12 source frames, configs[4] 1080p x 48 (dynamic-mask compositing)
        cfg_stats = {}
        for label, size in (("C2_960x540x12", (540, 960, 12)), ("C5_1920x1080x48", (1080, 1920, 48))):
            if size == (H, W, S):
                continue
            try:
                _, vs_, _, rv_ = load_scene("nominal", size)
                cfg_stats[label] = stats_of(rv_, vs_, label, arrangements=lane_candidates if auto_lanes else None)
                cfg_stats[label].update({"height": size[0], "width": size[1], "src_frames": size[2]})
                del vs_, rv_
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- the headline does not depend on it; the line says what failed
                cfg_stats[label] = {"error": f"{type(e).__name__}: {e}"}
        variants["configs"] = cfg_stats
        rvr.set_lanes(*lane_cfg)

    # ---------------- CPU baseline: the oracle (port of the reference algorithm) on host cores, on the SAME
    # workload (this video, view 0).  Aggregation and the dynamic branch (brute-force kNN as pytorch3d's) run in
    # full; the naive rasteriser -- every pixel scans every point, cost = pixels x points -- runs on pixel
    # windows holding ~1/32 of the frame with ALL points and is extrapolated by the pixel ratio (the law is
    # exact for a loop whose per-pixel cost does not depend on the pixel).  configs[0] (256 x 256 x 4) runs in
    # full as well, and the HIP renderer is checked against it through the evaluator-shaped harness.
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc

        cores = orc.num_threads()
        c0 = time.perf_counter()
        o_cloud = orc.aggregate_static_pcl(video["rgbs"], video["depths"], video["dyn_masks"], K3s, c2ws)
        t_agg = time.perf_counter() - c0
        assert o_cloud.shape[0] == n_static, (o_cloud.shape, n_static)  # same cloud as the HIP path (tests: bit-exact)
        v0 = synth.make_view(video, view_ids[0], frac=0.4, seed=5)
        c0 = time.perf_counter()
        ndc = orc.points_to_ndc(o_cloud[:, :3], v0["flat_cam_tgt"][0], H, W)
        wh, ww = max(8, H // 8), max(8, W // 16)
        wins = [(0, 0), (H // 2 - wh // 2, W // 2 - ww // 2), (H - wh, W - ww), (H // 4, (5 * W) // 8)]
        for (y0, x0) in wins:
            fr = orc.rasterize_points_window(ndc, H, W, float(rc.st_render_pcl_pt_radius), K, y0, y0 + wh, x0, x0 + ww)
            orc.composite(fr[0], fr[2], float(rc.st_render_pcl_pt_radius), o_cloud[:, 3:])
        t_win = time.perf_counter() - c0
        factor = (H * W) / float(len(wins) * wh * ww)
        od = dict(v0)
        od["rgb_gnt"] = np.zeros((1, H, W, 3), np.float32)  # static image supplied: times the dynamic branch + composite only
        c0 = time.perf_counter()
        orc.render_view(od, dict(rc), static_noise=v0["static_noise"], alpha=100.0)
        t_dyn = time.perf_counter() - c0
        t_full = t_agg + t_win * factor + t_dyn
        # configs[0] in full on the CPU, and the HIP renderer against it through the evaluator's step
        sv = synth.make_video(4, 256, 256, seed=1234)
        sd = synth.make_view(sv, 1, frac=0.4, seed=5)
        c0 = time.perf_counter()
        sc = orc.aggregate_static_pcl(sv["rgbs"], sv["depths"], sv["dyn_masks"], sv["K3s"], sv["c2ws"])
        od = dict(sd)
        od["st_pcl_rgb"] = sc[None]
        o1 = orc.render_view(od, dict(rc), static_noise=sd["static_noise"], alpha=100.0)
        t_c1 = time.perf_counter() - c0
        hd = {k: torch.from_numpy(np.ascontiguousarray(x)) for k, x in sd.items()}
        hd["st_pcl_rgb"] = ops.static_aggregate(T(sv["rgbs"]), T(sv["depths"]), T(sv["dyn_masks"]).view(torch.uint8), sv["K3s"], sv["c2ws"])[0][None, :sc.shape[0]].contiguous()
        hd["rgb_tgt"] = torch.from_numpy(np.ascontiguousarray(o1["combined_rgb"].transpose(0, 2, 3, 1)))
        hd["eval_mask"] = torch.from_numpy(np.repeat(o1["render_dyn_mask"].transpose(0, 2, 3, 1), 3, axis=-1).astype(np.float32))
        md, ex = harness.eval_step(model, hd, rc, device=dev, return_images=True)
        raw = ex["ret"]["combined_rgb"].cpu().numpy()
        mse = float(np.mean((raw.astype(np.float64) - o1["combined_rgb"]) ** 2))
        cpu_baseline = {
            "value": round(1.0 / t_full, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"this workload, one view ({W}x{H}x{S}, {n_static} pts): aggregation {t_agg:.1f} s + dyn branch (brute kNN) {t_dyn:.1f} s "
                      f"in full; naive rasteriser on {len(wins)} {ww}x{wh} windows {t_win:.1f} s, x{factor:.0f} by pixel ratio",
            "sample_detail": f"the GPU workload itself ({W}x{H}, {S} source frames, {n_static} static points): static aggregation "
                             f"({t_agg:.2f} s) and dynamic branch with brute-force kNN + composite ({t_dyn:.2f} s) in full; naive "
                             f"O(pixels x points) rasteriser as pytorch3d bin_size=0 on {len(wins)} windows of {ww}x{wh} pixels with all points "
                             f"({t_win:.2f} s), extrapolated x{factor:.1f} by the pixel ratio",
            "measured_seconds": round(t_agg + t_win + t_dyn, 2), "estimated_seconds_per_view": round(t_full, 2),
            "extrapolation": {"law": "naive rasteriser time = pixels x points x const; same points, pixel ratio", "factor": round(factor, 2),
                              "applies_to_seconds": round(t_win, 2)},
            "configs0_256x256x4_full": {"seconds_per_view": round(t_c1, 3), "frames_per_s": round(1.0 / t_c1, 3)},
            "hip_vs_oracle_configs0": {
                "through": "pgdvs_amd.harness.eval_step (evaluator_pgdvs.py:26-188), ground truth = the oracle's image",
                "psnr_full_quantised_db": float(md["eval/psnr_full_combined"]),
                "note_quantised": "0 = identical 8-bit images (the reference's calculate_psnr returns 0 for mse == 0)",
                "differing_8bit_values": int((ex["pred"] != ex["gt"]).sum()),
                "psnr_unquantised_db": round(10 * np.log10(1.0 / mse), 1) if mse > 0 else None,
                "max_abs_diff": float(np.abs(raw - o1["combined_rgb"]).max())}}

    # ---------------- GNT sub-benchmark (BASELINE configs[2]: "GNT feature aggregation on MFMA")
    # A full 1080p / 24-view / 256-sample GNT frame is ~3.3 PFLOP (>= 20 s even at the fp32-MFMA
    # peak), so -- as BASELINE.md section 3 prescribes -- it is reported on a ray subset as
    # TFLOP/s and is not part of `value` (whose static image comes from the point renderer).
    gnt = None
    if rank == 0 and world == 1 and args.gnt_rays > 0:
        from pgdvs_amd.models.gnt.models.transformer_network import GNT

        torch.manual_seed(0)
        net = GNT(netwidth=64, transformer_depth=8).to(dev).eval()
        Rg, Sg, Vg = args.gnt_rays, 256, S
        # A13 feeds A14 with REAL projections: a chunk of this view's target rays is sampled along the ray and
        # projected into the S resident source frames (pgdvs_gnt_gather: rgb + 32-channel feature rows, ray
        # differences, in-bounds / dynamic masks); the feature maps are random (the ResUNet is a torch/MIOpen row)
        g = torch.Generator(device=dev).manual_seed(1)
        v0 = views[0]
        cam_t = ops.cam_prep(v0["flat_cam_tgt"][0])
        cams_s = ops.cam_prep(torch.stack([torch.from_numpy(synth.flat_cam(H, W, K3s[i], c2ws[i])) for i in range(S)]).to(dev))
        ro, rd, _, _ = ops.get_rays(cam_t, H, W, 1)
        pick = torch.randperm(H * W, device=dev, generator=g)[:Rg].sort().values
        ro, rd = ro[pick].contiguous(), rd[pick].contiguous()
        featmaps = torch.randn(S, (H + 3) // 4, (W + 3) // 4, 32, device=dev, generator=g)
        inv_masks = T(video["dyn_masks"].astype(np.float32))
        drange = torch.tensor([[0.8, 5.0]], device=dev)

        def gnt_chunk():
            gg = ops.gnt_gather(ro, rd, drange, Sg, True, cam_t, cams_s, rvr.video["rgbs"], featmaps, inv_masks)
            out = net(gg["rgb_feat"], gg["ray_diff"], gg["mask"], gg["pts"], rd, ret_view_entropy=True, ret_view_std=True)
            return gg, out

        def telemetry():
            """board power (W) from sysfs while a chunk runs, or None (the clock level sysfs marks as current does not
            move on this driver and is no longer recorded)"""
            import glob
            try:
                for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
                    return {"power_W": round(int(open(f).read()) / 1e6, 1)}
            except Exception:  # noqa: BLE001 -- telemetry is optional
                pass
            return None

        with torch.no_grad():
            gg, out_k = gnt_chunk()
            torch.cuda.synchronize()
            valid_frac = float(gg["mask"].mean())
            # the timed network's output is CHECKED: 64 rays of the chunk through the torch statement (fused kernels off)
            ops._GNT_VIEW_ENABLED = False
            try:
                out_m = net(gg["rgb_feat"][:64], gg["ray_diff"][:64], gg["mask"][:64], gg["pts"][:64], rd[:64],
                            ret_view_entropy=True, ret_view_std=True)
            finally:
                ops._GNT_VIEW_ENABLED = True
            gnt_parity = {"rays": 64, "rgb_max_abs_diff": float((out_k[0][:64, :3] - out_m[0][:, :3]).abs().max()),
                          "weights_max_rel_diff": float(((out_k[0][:64, 3:] - out_m[0][:, 3:]).abs() / out_m[0][:, 3:].abs().clamp(min=1e-12)).max()),
                          "extras_max_abs_diff": max(float((out_k[1][k_][:64] - out_m[1][k_]).abs().max()) for k_ in out_m[1])}
            assert gnt_parity["rgb_max_abs_diff"] <= 1e-4 and gnt_parity["extras_max_abs_diff"] <= 1e-4, gnt_parity
            e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # ten repetitions, each timed on its own (wall clock around gather + transformer, HIP events for the split), with
            # the clock / power the driver reports right after each: the figure quoted is the MEDIAN, the spread is beside it
            reps, tele = [], []
            for _ in range(10):
                g0 = time.perf_counter()
                e0.record()
                gg = ops.gnt_gather(ro, rd, drange, Sg, True, cam_t, cams_s, rvr.video["rgbs"], featmaps, inv_masks)
                e1.record()
                net(gg["rgb_feat"], gg["ray_diff"], gg["mask"], gg["pts"], rd, ret_view_entropy=True, ret_view_std=True)
                e2.record()
                tele.append(telemetry())  # (read while the chunk is still running: after the synchronise the card idles)
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - g0, e0.elapsed_time(e1), e1.elapsed_time(e2)))
            # the same chunk with every product on the fp32 matrix instruction (library option gnt_fp32): five repetitions
            ops.set_option("gnt_fp32", 1)
            try:
                gnt_chunk()
                torch.cuda.synchronize()
                reps32 = []
                for _ in range(5):
                    g0 = time.perf_counter()
                    gnt_chunk()
                    torch.cuda.synchronize()
                    reps32.append(time.perf_counter() - g0)
            finally:
                ops.set_option("gnt_fp32", 0)
            reps32.sort()
        reps.sort()
        gdt, t_gather, t_net = reps[len(reps) // 2]
        gflop = 2.0 * Rg * Sg * (1048064 + 84416 * Vg)
        gather_bytes = Rg * Sg * Vg * (4 * 35 * 4 + (44 + 4 * 32))  # 4 bilinear corners x 35 channels read, one row written
        tf = lambda sec: round(gflop / sec / 1e12, 2)  # noqa: E731
        pw = [t_["power_W"] for t_ in tele if t_ and "power_W" in t_]
        gnt = {"rays": Rg, "samples_per_ray": Sg, "views": Vg, "layers": 8, "ms_per_chunk": round(gdt * 1e3, 2),
               "ms_gather_A13": round(t_gather, 3), "ms_transformer_A14": round(t_net, 3),
               "tflops": tf(gdt), "tflops_A14_alone": round(gflop / (t_net * 1e-3) / 1e12, 2),
               "repetitions": {"n": len(reps), "statistic": "median", "tflops_min": tf(reps[-1][0]), "tflops_max": tf(reps[0][0]),
                               "ms_per_chunk_all": [round(r_[0] * 1e3, 2) for r_ in reps],
                               "power_W_range": [min(pw), max(pw)] if pw else None},
               "peak_tflops_fp32_mfma": 157.3, "fp32_equivalent_frac": round(gflop / gdt / 157.3e12, 4),
               "mfma_busy_frac": gnt_pmc_profile()[1] or None,
               "mfma_busy_source": (f"profiles/{gnt_pmc_profile()[0]}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x duration) per kernel, "
                                    "committed rocprofv3 --pmc pass of tools/gnt_bench.py (1024 rays x 24 views x 256 samples); not measured in this run"),
               "fp32_instruction_path": {"ms_per_chunk": round(reps32[len(reps32) // 2] * 1e3, 2), "tflops": tf(reps32[len(reps32) // 2]),
                                         "fp32_equivalent_frac": round(gflop / reps32[len(reps32) // 2] / 157.3e12, 4), "repetitions": len(reps32),
                                         "note": "option gnt_fp32 (PGDVS_GNT_FP32=1 at load time): every product on the fp32 matrix instructions (v_mfma_f32_16x16x4_f32 / 32x32x2_f32)"},
               "gather_alg_GBps": round(gather_bytes / (t_gather * 1e-3) / 1e9, 1), "valid_projection_fraction": round(valid_frac, 3),
               "parity_vs_torch_statement": gnt_parity,
               "dtype": "f32 inputs, weights and results; the view layers' 64 x 64 products and the feed-forward blocks run as bf16x3 products on "
                        "v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 (both operands split exactly into three bf16 pieces, six partial products, "
                        "fp32 accumulation), everything else on v_mfma_f32_16x16x4_f32",
               "est_seconds_per_1080p_frame": round(gdt * (H * W / Rg), 1),
               "note": "pgdvs_gnt_gather (real projections of target-ray samples into the resident source frames, dynamic masks "
                       "applied) + GNT.forward incl. view entropy/std side outputs; FLOPs = the fp32 multiply-adds of A14 (an operand "
                       "split into bf16 pieces is not counted three times), time for both; fp32_equivalent_frac = those FLOPs against the "
                       "fp32-MFMA peak whichever instruction a product runs on -- NOT a utilisation of what is issued (six bf16 MFMAs at 16x "
                       "the fp32 rate put the ceiling of a split product at ~2.7x that peak): the matrix pipe's own busy fraction per kernel "
                       "is mfma_busy_frac; "
                       "median of ten repetitions (board power as sysfs reports it while each runs)"}
        # The whole renderer with the GNT static renderer at the reference's own benchmark setting
        # (NVIDIA Dynamic Scenes: 288 x 550 targets, 10 spatial + 2 temporal source views, 256 samples
        # per ray, chunks of 1024 rays).  Informational, never `value`.
        try:
            gnt["pgdvs_forward_288x550_10views"] = gnt_full_frame(dev)
        except Exception as e:  # noqa: BLE001 -- the headline number does not depend on it
            gnt["pgdvs_forward_288x550_10views"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        frames = args.steps * world
        fps = frames / elapsed
        alg_total = (20 * S + 120) * H * W
        out = {
            "metric": "novel-view frames/s at 1080p x 24 src frames; achieved HBM GB/s vs gfx950 peak",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "latency_ms": latency_ms, "eval_step_frames_per_s": (eval_loop or {}).get("frames_per_s"), "eval_step": eval_loop,
            "host_enqueue_ms_per_step": round(host_ms, 3), "host_blocked_on_gpu_ms_per_step": round(host_wait_ms, 3),
            "host_native_call_ms_per_step": round(host_native_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{W}x{H} target view, {S} source frames resident in HBM: static aggregation (A12) + "
                            f"point z-buffer raster K={K} (A9) + flow-warped dynamic splat (A1-A8, outlier filter "
                            f"{'on' if not args.no_outlier else 'off'}) + composite (A11)",
                "scene": args.scene, "views_in_flight": n_lanes, "views_in_flight_choice": lanes_note, "host_run_ahead_views": args.run_ahead, "memory": mem_note, "raster_row_bound": rvr.row_bound,
                "launch": ("per-op: ~85 C-ABI calls per view enqueued from Python (--per-op)" if args.per_op else
                           "one native call per view (pgdvs_view_geo_forward: A12 + A9 + A2-A8 + A11 enqueued from C++)")
                          + (", dynamic-branch geometry on a second stream per lane" if rvr.side_streams else "")
                          + (", lane streams picked by hardware queue" if rvr.place_streams else ""), "stream_queue_groups": rvr.queue_groups, "height": H, "width": W, "src_frames": S, "static_points": n_static, "dyn_pixels": n_dyn,
                "parallelism": f"frames sharded over {world} GPU(s), RCCL gather of the image stack" if world > 1 else "1 GPU",
                "per_rank_frames_per_s": [round(args.steps / x, 2) for x in per_rank_s],
                "gather_bytes_to_rank0": int(3 * H * W * 4 * args.steps * (world - 1)),
                "gather_GBps_into_rank0": round(3 * H * W * 4 * args.steps * (world - 1) / elapsed / 1e9, 2),
                "gather_receive_ring_slots": ring_slots,
                "collective_backend": (dist.get_backend() if world > 1 else None), "rccl_ranks_seen": [p_[0] for p_ in peers],
                "rank_devices": [f"rank {p_[0]}: cuda:{p_[1]} {p_[2]}" for p_ in peers],
                "stream_placement_per_rank": [p_[3] for p_ in peers], "stream_queue_probe": rvr.queue_probe,
                "whole_view_alg_bytes": alg_total,
                "whole_view_alg_GBps": round(alg_total * fps / 1e9 / max(world, 1), 2),
            },
            "steady_state": steady,
            "roofline": roofline, "roofline_kernels": roofline_kernels,
            # BASELINE.md section 3 defines the path's roofline figure over the whole view:
            # bytes(S,P) = (20 S + 120) H W algorithmic bytes per novel view x views per second per GPU
            "roofline_path": {"bound": "hbm", "achieved": round(alg_total * fps / 1e9 / max(world, 1), 2), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(alg_total * fps / 1e9 / max(world, 1) / HBM_PEAK_GBS, 5),
                              "alg_bytes_per_view": alg_total, "traffic_bytes_per_view": measured_view_traffic(H, W, S),
                              "traffic_source": "committed profile (see roofline.traffic_source)",
                              "note": "all kernels of a view; the path is bound by search / z-buffer / fp64 re-projection work, "
                                      "not by streaming its inputs (DESIGN.md section 4)"},
            "cpu_baseline": cpu_baseline, "gnt": gnt, "variants": variants, "kernels": kernels,
        }
        emit(out)
    if world > 1:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
