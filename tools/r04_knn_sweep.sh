#!/bin/bash
# round 4: the kNN grid's density knob (PGDVS_KNN_PER_CELL: results identical for any value) against the thread-per-query
# pass and the ring search it feeds; repeatability of 3 lanes with / without the second stream
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --inflight 3"
show() { python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.loads([l for l in open(f"gpurun_out/r04/{n}.json") if l.startswith("{")][-1])
    k=b["kernels"]
    print(n, "value", b["value"], "steady", (b.get("steady_state") or {}).get("frames_per_s"), "lat", b["latency_ms"]["median"],
          "tpq", k.get("grid_query_tpq",{}).get("ms_per_step"), "ring", k.get("grid_query",{}).get("ms_per_step"), "coarse", k.get("grid2_query",{}).get("ms_per_step"))
except Exception as e:
    print(n, "unparsed:", e)
PY
}
for pc in 10 14 18 24 32; do
  PGDVS_KNN_PER_CELL=$pc $B > gpurun_out/r04/knn_pc$pc.json 2> gpurun_out/r04/knn_pc$pc.err; show knn_pc$pc
done
for r in 1 2; do
  $B > gpurun_out/r04/rep_l3_$r.json 2>/dev/null; show rep_l3_$r
  $B --side-stream > gpurun_out/r04/rep_side3_$r.json 2>/dev/null; show rep_side3_$r
done
