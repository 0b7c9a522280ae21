#!/bin/bash
# tools/ab_rounds.sh [N [STEPS]] -- on the GPU box: this tree's bench line and an older tree's (exported to build/r03tree by
# `git archive <rev> | tar -x -C build/r03tree` and built there), alternating, in the driver's form (--steps $STEPS --warmup 5).
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-5}; STEPS=${2:-20}
one() {
  (cd "$1" && python3 bench.py --steps $STEPS --warmup 5 --cpu-frames 0 --orbit-frames 0 --dropin-frames 0 --frames-per-launch 1 --no-verify 2>/dev/null | tail -1) | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$2', 'steps', j['steps'], j['ms_per_step'], 'region', r.get('kernel_ms_avg'), flush=True)"
}
for i in $(seq $N); do one "$R/build/r03tree" r03; one "$R" now; done
