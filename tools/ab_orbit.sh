#!/bin/bash
# tools/ab_orbit.sh [LIB...] -- on the GPU box: standing frame, orbit leg and the orbit with 4 cameras per launch for the in-tree library and the given A/B builds, two rounds
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
for rep in 1 2; do for lib in "" "$@"; do
  RTO_HIP_LIB=${lib:+$R/$lib} python3 bench.py --steps 400 --warmup 20 --cpu-frames 0 --dropin-frames 0 --no-verify --orbit-frames 240 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); o=j.get('orbit') or {}
print('%-36s static %.5f orbit %.5f, 4 cameras per launch %.5f; standing, 4 per launch %.5f' % ('${lib:-in-tree}', j['ms_per_step'], o.get('ms_per_frame', 0), (o.get('frames_per_launch') or {}).get('ms_per_frame', 0), (j.get('frames_per_launch') or {}).get('ms_per_frame', 0)), flush=True)"
done; done
