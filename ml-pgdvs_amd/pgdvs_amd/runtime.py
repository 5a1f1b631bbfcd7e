"""Resident-video rendering: the arrangement ``bench.py`` times, ``harness.eval_step`` can be pointed at and the C3 parity
test renders (`ResidentVideoRenderer`).

(Rounds 2-4 also carried ``GraphedRender``, a HIP-graph replay of a whole view.  With one native call per view the host
needs 0.25 ms per view and the replay path was never faster than eager launches -- 800-1017 against 845-1075 frames/s in
rounds 2-3 -- and on this ROCm a replay right behind a kernel on the NULL stream ended in a memory fault;
removed in round 5.)
"""
import torch


def stream_queue_groups(streams, cycles: int = 40000, links: int = 10):
    """Which of ``streams`` share a HARDWARE queue?  HIP multiplexes its streams onto a few hardware queues (four by default;
    more do not run side by side either: tools/overlap_probe.py), and two streams on one queue run one behind the other
    however idle the chip is -- a lane whose stream lands on the queue of another lane's aggregation chain waits for that
    chain.  The mapping is the runtime's (creation / first-use order) and cannot be queried, but it shows: two chains of
    dependent single-workgroup spin kernels (``torch.cuda._sleep``) take the time of one when their streams have queues of
    their own and of two when they share one.  Returns a list of groups (lists of indices into ``streams``); ~1 ms per
    comparison, every stream is compared with one member of each group found so far."""
    import time

    def chains(ss):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in ss:
            with torch.cuda.stream(st):
                for _ in range(links):
                    torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    chains([streams[0]])  # (first launches: lazy queue creation)
    single = min(chains([streams[0]]) for _ in range(3))
    groups, ratios = [], []
    for i, st in enumerate(streams):
        for g in groups:
            # one behind the other: ~2 x single; side by side: ~1.15 x.  The clock is the host's: a ratio between 1.35 and 1.8
            # (a busy host, a clock step) is measured again, and a comparison that stays there makes the whole probe fail --
            # the caller then takes creation order (a wrong grouping would silently cost a quarter of the throughput)
            for attempt in range(4):
                r = min(chains([streams[g[0]], st]) for _ in range(2)) / single
                if not 1.35 < r < 1.8:
                    break
                single = min(single, min(chains([streams[0]]) for _ in range(2)))
            else:
                LAST_PROBE.update({"ratios": ratios + [round(r, 2)], "ok": False})
                raise RuntimeError(f"stream_queue_groups: ratio {r:.2f} between 'side by side' and 'one behind the other'")
            ratios.append(round(r, 2))
            if r >= 1.8:
                g.append(i)
                break
        else:
            groups.append([i])
    LAST_PROBE.update({"ratios": ratios, "ok": True})
    return groups


def stream_on_other_queue(main, candidates: int = 5, cycles: int = 40000, links: int = 8):
    """A new stream that does NOT share ``main``'s hardware queue (the renderer's default second stream: on the queue of the
    stream it is forked from, the dynamic branch would run behind the static one instead of beside it -- 1.29 ms per view
    instead of 1.02 in round 6's evaluator-shaped loop, depending on how many streams the process had created before).
    Up to ``candidates`` fresh streams are tried with the pair test of ``stream_queue_groups``; the first one that runs side
    by side with ``main`` is returned, the first one created if none does (or if the clock is too noisy to tell)."""
    import time

    def chains(ss):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in ss:
            with torch.cuda.stream(st):
                for _ in range(links):
                    torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dev = main.device
    first = None
    try:
        chains([main])
        single = min(chains([main]) for _ in range(3))
        for _ in range(candidates):
            st = torch.cuda.Stream(device=dev)
            first = first or st
            if min(chains([main, st]) for _ in range(2)) < 1.35 * single:
                return st
    except Exception:  # noqa: BLE001 -- an optimisation: any stream is correct
        pass
    return first or torch.cuda.Stream(device=dev)


LAST_PROBE = {}  # the last probe's pair ratios (diagnostics: ResidentVideoRenderer.queue_probe, bench.py's detail record)


class ResidentVideoRenderer:
    """Novel views of ONE video whose S source frames stay resident in HBM, rendered through ``PGDVSRenderer.forward``
    with the static cloud aggregated per view (A12) inside the same native call (``data["_st_pcl_video"]``): what
    ``bench.py`` times, what ``harness.eval_step`` can be pointed at, and what the C3 parity test renders -- one
    arrangement for all three.

    ``lanes`` independent views may be in flight, each on its own HIP stream (``side_streams``: plus a second stream per
    lane for the dynamic branch's geometry, forked and joined inside the native call).  Cloud buffers and the
    rasteriser's tile lists are capacity-sized (S*H*W rows) until ``calibrate`` has read one view's count back; after
    that they are bounded by 1.25 x that count + 65536 rows.  A view that outgrows the bound is TRUNCATED by the
    aggregation, which clamps its count at the buffer's rows -- the rasteriser's status word cannot fire for it (it only
    sees the clamped count): the one signal is a count equal to the bound.  ``check_overflow(ret)`` tests exactly that
    (one host read; ``harness.eval_step`` makes the same test with the words it reads back anyway).
    """

    def __init__(self, model, render_cfg, rgbs, depths, dyn_masks, K3s, c2ws, *, lanes: int = 3, side_streams: bool = False,
                 native: bool = True, place_streams: bool = False):
        import numpy as np

        self.model, self.rc = model, render_cfg
        self.dev = rgbs.device
        S, H, W = depths.shape
        self.S, self.H, self.W = S, H, W
        self.capacity = S * H * W
        self.row_bound = None
        self.native = native
        self.video = {"rgbs": rgbs.contiguous(), "depths": depths.contiguous(),
                      "dyn_masks": (dyn_masks.view(torch.uint8) if dyn_masks.dtype == torch.bool else dyn_masks).contiguous(),
                      "K3s": np.ascontiguousarray(K3s, dtype=np.float64), "c2ws": np.ascontiguousarray(c2ws, dtype=np.float64)}
        self.side_streams = side_streams
        self.place_streams = place_streams  # lane streams picked by hardware queue (see _place)
        self.queue_groups = None
        self.queue_probe = None
        self.set_lanes(lanes)

    def set_lanes(self, n: int, side_streams=None, place_streams=None) -> None:
        """``n`` views in flight; optionally another arrangement of the lanes' streams (second stream per lane or not, streams
        picked by hardware queue or in creation order): every arrangement keeps its own streams, so switching back and forth
        (bench.py's probe) reuses them."""
        if side_streams is not None:
            self.side_streams = bool(side_streams)
        if place_streams is not None:
            self.place_streams = bool(place_streams)
        sets = self.__dict__.setdefault("_lane_sets", {})
        key = (self.side_streams, self.place_streams)
        self.lanes = sets.setdefault(key, [])
        # (default priorities: raising the main or the side streams' costs a quarter of the throughput, round 5)
        if len(self.lanes) < n and self.place_streams:
            self._place(n)
            sets[key] = self.lanes
        while len(self.lanes) < n:
            self.lanes.append((torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev) if self.side_streams else None))
        self.n_lanes = n

    def drop_other_lane_sets(self) -> None:
        """keep the current arrangement's lanes only (after a probe of several): the other arrangements' streams and the
        model's workspaces on them are released"""
        torch.cuda.synchronize(self.dev)
        sets = self.__dict__.get("_lane_sets", {})
        key = (self.side_streams, self.place_streams)
        for k in [k_ for k_ in sets if k_ != key]:
            sets.pop(k)
        del self.lanes[self.n_lanes:]
        if hasattr(self.model, "release_view_states"):
            self.model.release_view_states([st for pair in self.lanes for st in pair])
        torch.cuda.empty_cache()

    def _place(self, n: int) -> None:
        """lane streams chosen by HARDWARE QUEUE (stream_queue_groups): every lane's main stream -- it carries the view's
        longest dependent chains: the aggregation's links, the rasteriser -- on a queue of its own as far as the queues
        go, the second streams (dynamic branch) together on what is left, so that no lane's static branch ever queues
        behind another lane's.  Falls back to plain creation order when fewer than two queues show."""
        pool = [torch.cuda.Stream(device=self.dev) for _ in range(max(12, 3 * n))]
        try:
            with torch.cuda.device(self.dev):
                groups = stream_queue_groups(pool)
        except Exception:  # noqa: BLE001 -- placement is an optimisation: without the probe the lanes take creation order
            self.queue_groups = None
            self.queue_probe = dict(LAST_PROBE, fallback="creation order")
            return
        self.queue_groups = [len(g) for g in groups]
        self.queue_probe = dict(LAST_PROBE)
        if len(groups) < 2:
            return
        groups.sort(key=len, reverse=True)
        n_main_q = min(n, len(groups) - (1 if self.side_streams and len(groups) > n else 0)) or 1
        main_q, side_q = groups[:n_main_q], (groups[n_main_q:] or groups[-1:])
        lanes = []
        used = set()
        for i in range(n):
            g = main_q[i % len(main_q)]
            mi = next((k for k in g if k not in used), g[0])
            used.add(mi)
            side = None
            if self.side_streams:
                gs = side_q[i % len(side_q)]
                si = next((k for k in gs if k not in used), gs[0])
                used.add(si)
                side = pool[si]
            lanes.append((pool[mi], side))
        replaced = {id(st) for pair in self.lanes for st in pair if st is not None}
        self.lanes = lanes
        if replaced and hasattr(self.model, "release_view_states"):
            # (growing a placed arrangement replaces its streams: the native call's workspaces on the old ones -- 1-2 GB each at
            # 1080p -- would otherwise stay until the renderer's LRU bound evicts them)
            torch.cuda.synchronize(self.dev)
            sets = self.__dict__.get("_lane_sets", {})
            keep = [st for ls in sets.values() for pair in ls for st in pair if st is not None and id(st) not in replaced]
            keep += [st for pair in lanes for st in pair if st is not None]
            self.model.release_view_states(keep)

    def calibrate(self, data) -> int:
        """render one view with capacity-sized buffers, read its count back (one host synchronisation) and bound the
        buffers of the views that follow; returns the count"""
        from . import ops

        self.row_bound = None
        ret, main = self.render(data, 0)
        torch.cuda.current_stream(self.dev).wait_stream(main)
        n = ops.checked_count(ret["st_pcl_rgb_count"], "pgdvs_static_aggregate")
        self.row_bound = min(self.capacity, int(1.25 * n) + 65536)
        return n

    def render(self, data, lane: int, out=None, use_side: bool = True):
        """enqueue one view on lane ``lane``; ``out`` [1,3,H,W]: the caller's slot for ``combined_rgb`` (written by the
        splat epilogue itself); ``use_side=False``: everything on the lane's main stream (per-kernel timing: no kernel of
        the view then runs beside another).  Returns (ret dict incl. ``st_pcl_rgb`` / ``st_pcl_rgb_count``, the lane's stream)."""
        from . import ops

        main, side = self.lanes[lane % self.n_lanes]
        if not use_side:
            side = None
        main.wait_stream(torch.cuda.current_stream(self.dev))
        d = dict(data)
        if out is not None:
            d["_combined_rgb_out"] = out
        cap = self.row_bound or self.capacity
        with torch.cuda.stream(main), torch.no_grad():
            if self.native:
                v = dict(self.video)
                v["capacity"] = cap
                d["_st_pcl_video"] = v
                if self.row_bound is not None:
                    d["st_pcl_rgb_row_bound"] = self.row_bound
                d["_side_stream"] = side if side is not None else False  # (False: everything on the lane's stream)
                ret = self.model.forward(d, render_cfg=self.rc, disable_tqdm=True)
            else:
                # the per-op arrangement of rounds 1-3: ~85 C-ABI calls enqueued from Python
                d["_dyn_prepared"] = self.model.dyn_renderer.prepare(d, self.rc, stream=side)
                v = self.video
                cloud, cnt, xyz = ops.static_aggregate(v["rgbs"], v["depths"], v["dyn_masks"], v["K3s"], v["c2ws"], capacity=cap,
                                                       return_xyz=True)
                d["st_pcl_rgb"], d["st_pcl_rgb_count"], d["st_pcl_xyz"] = cloud[None], cnt, xyz[None]
                if self.row_bound is not None:
                    d["st_pcl_rgb_row_bound"] = self.row_bound
                ret = self.model.forward(d, render_cfg=self.rc, disable_tqdm=True)
                ret["st_pcl_rgb"], ret["st_pcl_rgb_count"], ret["st_pcl_xyz"] = cloud[None], cnt, xyz[None]
        return ret, main

    def check_overflow(self, ret) -> int:
        """the view's cloud count (one host synchronisation); raises when the aggregation reported an error or the cloud
        filled the bounded buffer (rows may have been dropped: render with a larger bound / ``calibrate`` again)"""
        from . import ops

        n = ops.checked_count(ret["st_pcl_rgb_count"], "pgdvs_static_aggregate")
        if self.row_bound is not None and self.row_bound < self.capacity and n >= self.row_bound:
            raise ops.PgdvsHipError(f"the aggregated static cloud filled its bounded buffer of {self.row_bound} rows: rows may have "
                                    "been dropped -- calibrate() on this view or raise row_bound")
        ops.check_raster_status(ret.get("geo_static_raster_status", None))
        return n

    def join(self) -> None:
        cur = torch.cuda.current_stream(self.dev)
        for main, side in self.lanes:
            cur.wait_stream(main)
