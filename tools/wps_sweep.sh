#!/bin/bash
# tools/wps_sweep.sh -- resident waves per SIMD of the lean octree launch (RTO_WAVES_PER_SIMD, read by -DRTO_DEV_KNOBS builds only:
# tools/build_variants.sh knobs ""), static frame + orbit leg, configs 2 and 4.  On the GPU box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for w in ${WPS:-4 6 0}; do
  for cfg in 2 4; do
    out=$(RTO_WAVES_PER_SIMD=$w RTO_HIP_LIB=$PWD/build/variants/librto_hip_knobs.so python3 bench.py --config $cfg --no-extras --cpu-frames 0 --orbit-frames 120 --steps 200 --warmup 20 2>&1 | tail -1)
    case "$out" in *"GPU core dump"*|*"Memory access fault"*) echo "FAULT $out"; exit 1;; esac
    python3 -c "
import json,sys
j=json.loads(sys.argv[1]); o=j.get('orbit',{}); print('w$w cfg$cfg', round(j['ms_per_step']*1e3,2), 'us', j['verified_against_oracle'], 'orbit', {k:o[k] for k in o if 'ms' in k and not isinstance(o[k],dict)})" "$out"
  done
done
done
