#!/bin/bash
for n in 10 20 40 80; do
  python bench.py --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep --steps $n $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
