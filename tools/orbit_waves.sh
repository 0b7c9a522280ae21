#!/bin/bash
# tools/orbit_waves.sh -- on the GPU box, dev build: static frame and orbit leg at 4 / 5 / 6 / 8 resident waves per SIMD (RTO_WAVES_PER_SIMD)
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
export RTO_HIP_LIB=$R/build/variants/librto_hip_dev.so
for rep in 1 2; do for w in 4 5 6 8; do
  RTO_WAVES_PER_SIMD=$w python3 bench.py --steps 400 --warmup 20 --cpu-frames 0 --dropin-frames 0 --frames-per-launch 1 --no-verify --orbit-frames 240 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); o=j.get('orbit') or {}
print('waves per SIMD $w: static', j['ms_per_step'], 'orbit', o.get('ms_per_frame'), flush=True)"
done; done
