#!/bin/bash
# tools/occ_ab.sh -- A/B of the resident-waves knob (RTO_WAVES_PER_SIMD) with and without the occupancy mask; run on the GPU box.
for w in 0 4 5; do
  for m in "" "--no-tile-mask"; do
    out=$(RTO_WAVES_PER_SIMD=$w python3 bench.py --cpu-frames 0 --no-verify --orbit-frames 0 --dropin-frames 0 --steps 200 --warmup 20 $m "$@" 2>/dev/null | tail -1)
    python3 - "$w" "$m" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3]); r = d["roofline"]
print(f"waves/SIMD {sys.argv[1]:>2s} {sys.argv[2] or 'mask':15s} ms/frame {d['ms_per_step']:.5f}  kernel avg {r['kernel_ms_avg']:.5f} min {r['kernel_ms_event_pair_min']:.5f}  4-per-launch {d['frames_per_launch']['ms_per_frame']:.5f}")
PY
  done
done
