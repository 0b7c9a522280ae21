#!/bin/bash
# tools/ab_driver_form.sh -- the driver's bench form (--steps 20 --warmup 5) three times each with the mask on / off, on one box.
for rep in 1 2 3; do
  for m in "" "--no-tile-mask"; do
    out=$(python3 bench.py --cpu-frames 0 --no-verify --orbit-frames 0 --steps 20 --warmup 5 $m "$@" 2>/dev/null | tail -1)
    python3 - "$m" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print(f"{sys.argv[1] or 'mask':15s} ms/frame {d['ms_per_step']:.5f}  kernel avg {r['kernel_ms_avg']:.5f} med {r['kernel_ms_event_pair_median']:.5f} min {r['kernel_ms_event_pair_min']:.5f}  4-per-launch {d['frames_per_launch']['ms_per_frame']:.5f} dropin {d['dropin_call']['ms_per_call']:.5f} / {d['dropin_call']['ms_per_call_without_update']:.5f}")
PY
  done
done
