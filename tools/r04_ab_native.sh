#!/bin/bash
# round 4: the one-native-call-per-view path against the per-op arrangement, lanes and stream pairs
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep"
run() { name=$1; shift; echo "== $name: $*"; "$@" > gpurun_out/r04/$name.json 2> gpurun_out/r04/$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04/$name.err; }
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    b=json.loads([l for l in open(f"gpurun_out/r04/{n}.json") if l.startswith("{")][-1])
    print(n, "value", b["value"], "steady", (b.get("steady_state") or {}).get("frames_per_s"), "lat", b["latency_ms"]["median"], "host", b["host_enqueue_ms_per_step"], "native", b.get("host_native_call_ms_per_step"), "lanes", b["config"]["views_in_flight_choice"], "eval", b.get("eval_step_frames_per_s"), "mem", b["config"]["memory"]["reserved_GB_peak"])
except Exception as e:
    print(n, "unparsed:", e)
PY
}
run native_auto $B
run perop_auto $B --per-op
run native_side $B --side-stream
for k in 1 2 3 4 5 7; do run native_l$k $B --inflight $k --no-kernel-timing; done
for k in 2 3; do run side_l$k $B --inflight $k --side-stream --no-kernel-timing; done
run perop_l7 $B --per-op --inflight 7 --no-kernel-timing
