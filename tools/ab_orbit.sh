#!/bin/bash
# tools/ab_orbit.sh -- on the GPU box: static frame and orbit leg of the bench for the in-tree library and every A/B build under build/variants/
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
run() {
  RTO_HIP_LIB=$2 python3 bench.py --steps 400 --warmup 20 --cpu-frames 0 --dropin-frames 0 --frames-per-launch 1 --orbit-frames 240 2>&1 | tail -1 | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    j=json.loads(t); o=j.get('orbit') or {}
    print('%-10s static %.5f orbit %.5f verified=%s' % ('$1', j['ms_per_step'], o.get('ms_per_frame', 0), j.get('verified_against_oracle')), flush=True)
except Exception: print('$1 FAILED', t[-300:])"
}
for rep in 1 2; do
run base ""
for f in build/variants/librto_hip_*.so; do n=$(basename "$f" .so); run "${n#librto_hip_}" "$R/$f"; done
done
