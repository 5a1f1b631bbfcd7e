#!/bin/bash
# round 4, review item 1c: dense speculative depth loads in the aggregation's links (PGDVS_AGG_SPEC_DEPTH=1) against the gathers
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu -k "static_aggregation_shapes_vs_oracle or config_c1 or degenerate" 2>&1 | tail -2
PGDVS_AGG_SPEC_DEPTH=1 python -m pytest tests -x -q -m gpu -k "static_aggregation_shapes_vs_oracle or config_c1 or degenerate or config_c3_1080p" 2>&1 | tail -2
for r in 1 2; do
  bash tools/r04_kern.sh spec_off_$r | cut -c1-140
  bash tools/r04_kern.sh spec_on_$r PGDVS_AGG_SPEC_DEPTH=1 | cut -c1-140
done
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight 3"
for r in 1 2 3; do for v in 0 1; do echo -n "spec=$v: "; PGDVS_AGG_SPEC_DEPTH=$v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['frames_per_s'], d['latency_ms']['median'])"; done; done
