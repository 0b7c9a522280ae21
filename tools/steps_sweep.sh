#!/bin/bash
# tools/steps_sweep.sh -- ms/frame of the N=1 timed region as a function of --steps (graph replay and plain launches), one box.
for st in 20 20 40 100 200 1000; do
  for g in "" "--graph-frames 0"; do
    out=$(python3 bench.py --cpu-frames 0 --no-verify --orbit-frames 0 --dropin-frames 0 --frames-per-launch 1 --steps $st --warmup 5 $g "$@" 2>/dev/null | tail -1)
    python3 - "$st" "$g" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3]); r = d["roofline"]
print(f"steps {sys.argv[1]:>5s} {sys.argv[2] or 'graph':18s} ms/frame {d['ms_per_step']:.5f}  region gpu {r['timed_region_gpu_ms_per_frame']:.5f} kernel avg {r['kernel_ms_avg']:.5f} min {r['kernel_ms_event_pair_min']:.5f}")
PY
  done
done
