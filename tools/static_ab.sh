#!/bin/bash
# tools/static_ab.sh [LIB...] -- on the GPU box: the standing camera's frame (2000 frames, plain launches) for the in-tree library and the
# given A/B builds, alternating, four rounds
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
for rep in 1 2 3 4; do for lib in "" "$@"; do
  RTO_HIP_LIB=${lib:+$R/$lib} python3 bench.py --steps 2000 --warmup 200 --cpu-frames 0 --dropin-frames 0 --frames-per-launch 1 --no-verify --orbit-frames 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('%-40s static' % ('${lib:-in-tree}'), j['ms_per_step'], 'region', j['roofline']['kernel_ms_avg'], flush=True)"
done; done
