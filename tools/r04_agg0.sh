#!/bin/bash
# round 4: frame 0 of the aggregation as one fused launch (default) against the select + push pair (PGDVS_AGG_SPLIT0=1)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu -k "aggregat or config_c or agg or native or scene" 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --inflight 3"
show() { python - "$1" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.loads([l for l in open(f"gpurun_out/r04/{n}.json") if l.startswith("{")][-1])
    k=b["kernels"]
    print(n, "value", b["value"], "steady", (b.get("steady_state") or {}).get("frames_per_s"), "lat", b["latency_ms"]["median"],
          {x: round(k[x]["ms_per_step"]*1e3,1) for x in ("agg_frame0","agg_select","agg_push0","agg_step","agg_rows","grid_query_tpq","grid_query") if x in k},
          "sum", round(sum(v["ms_per_step"] for v in k.values())*1e3))
except Exception as e:
    print(n, "unparsed:", e); print(open(f"gpurun_out/r04/{n}.err").read()[-800:])
PY
}
for r in 1 2; do
  $B > gpurun_out/r04/agg0_fused_$r.json 2> gpurun_out/r04/agg0_fused_$r.err; show agg0_fused_$r
  PGDVS_AGG_SPLIT0=1 $B > gpurun_out/r04/agg0_split_$r.json 2> gpurun_out/r04/agg0_split_$r.err; show agg0_split_$r
done
