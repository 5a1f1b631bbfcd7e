#!/bin/bash
# round 4: isolated kernel times (one view in flight, one stream) under environment switches: `bash tools/r04_kern.sh TAG [ENV=VAL ...]`
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
tag=$1; shift
# BENCH_ARGS: extra flags for bench.py (e.g. "--scene noisy_depth")
env "$@" python bench.py ${BENCH_ARGS:-} --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep > gpurun_out/r04/kern_$tag.json 2> gpurun_out/r04/kern_$tag.err
python - "$tag" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.loads([l for l in open(f"gpurun_out/r04/kern_{n}.json") if l.startswith("{")][-1])
    k=b["kernels"]
    big={x: round(v["ms_per_step"]*1e3,1) for x,v in sorted(k.items(), key=lambda kv:-kv[1]["ms_per_step"]) if v["ms_per_step"]>=0.01}
    print(n, "lat", b["latency_ms"]["median"], "sum", round(sum(v["ms_per_step"] for v in k.values())*1e3), "launches", round(sum(v["launches_per_step"] for v in k.values())), big)
except Exception as e:
    print(n, "unparsed:", e); print(open(f"gpurun_out/r04/kern_{n}.err").read()[-800:])
PY
