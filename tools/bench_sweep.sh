#!/bin/bash
# bench.py over views-in-flight settings (GPU box); output: gpurun_out/sweep_inflight.log
for n in 1 2 3 4 6; do
  python bench.py --inflight $n --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep --steps 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
