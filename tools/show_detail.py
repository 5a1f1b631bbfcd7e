#!/usr/bin/env python3
"""print the per-kernel table and the variants of a bench detail record (gpurun_out/bench_detail.json or a path)"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_detail.json"))
ks = d["kernels"]
tot = 0
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if v["ms_per_step"] * 1e3 >= float(sys.argv[2] if len(sys.argv) > 2 else 0):
        print(f"{k:28s} n={v['launches_per_step']:5.1f} avg={v['avg_ms']*1e3:8.2f}us  per view={v['ms_per_step']*1e3:8.1f}us")
    tot += v["ms_per_step"]
print(f"sum {tot:.4f} ms, launches {sum(v['launches_per_step'] for v in ks.values()):.0f}")
print("value", d["value"], "steady", (d.get("steady_state") or {}).get("frames_per_s"), "latency", d["latency_ms"]["median"], "eval_step", d["eval_step_frames_per_s"],
      "host ms", d["host_enqueue_ms_per_step"], d["host_native_call_ms_per_step"])
print("lanes:", d["config"]["views_in_flight_choice"])
v = d.get("variants") or {}
print("cached cloud:", (v.get("static_cloud_aggregated_once_per_scene") or {}).get("frames_per_s"))
for k, o in (v.get("scenes") or {}).items():
    print("scene", k, o.get("frames_per_s"), o.get("error"))
for k, o in (v.get("configs") or {}).items():
    print("config", k, o.get("frames_per_s"), o.get("ms_per_view_runs"), (o.get("arrangement") or {}).get("ms_per_view_tried"), o.get("error"))
if d.get("eval_step"):
    print("eval_step", d["eval_step"]["host_ms_per_view"], d["eval_step"]["forward_and_synchronize_ms_per_view"])
