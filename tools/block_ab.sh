#!/bin/bash
# tools/block_ab.sh -- on the GPU box: workgroup size of the lean kernels (RTO_LEAN_BLOCK) x config
for cfg in 5 2 4; do
  for b in 256 128 64; do
    steps=400; [ $cfg = 5 ] && steps=100
    out=$(RTO_LEAN_BLOCK=$b python3 bench.py --config $cfg --steps $steps --warmup 20 --cpu-frames 0 --orbit-frames 0 --dropin-frames 0 2>/dev/null | tail -1)
    python3 - "$cfg" "$b" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[3])
    print(f"config {sys.argv[1]} block {sys.argv[2]:>3s}: {j['ms_per_step']*1e3:8.2f} us/frame  kernel avg {j['roofline']['kernel_ms_avg']*1e3:8.2f}  fpl4 {j.get('frames_per_launch',{}).get('ms_per_frame',0)*1e3:8.2f}  verified={j['verified_against_oracle']}")
except Exception as e:
    print("FAILED", sys.argv[1], sys.argv[2], sys.argv[3][-200:])
PY
  done
done
