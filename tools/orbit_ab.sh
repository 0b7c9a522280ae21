#!/bin/bash
# tools/orbit_ab.sh [PERIOD...] -- on the GPU box, dev build (tools/build_variants.sh dev ""): the bench's standing frame and its moving-camera
# leg (`orbit`) with the launch-order table of a moving camera rebuilt every PERIOD-th frame (RTO_ORDER_MOVED_PERIOD), two rounds
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
export RTO_HIP_LIB=$R/build/variants/librto_hip_dev.so
for rep in 1 2; do for p in ${@:-1 2 4 8}; do
  RTO_ORDER_MOVED_PERIOD=$p python3 bench.py --steps 400 --warmup 20 --cpu-frames 0 --dropin-frames 0 --frames-per-launch 1 --no-verify --orbit-frames 240 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); o=j.get('orbit') or {}
print('period $p: static', j['ms_per_step'], 'orbit', o.get('ms_per_frame'), flush=True)"
done; done
