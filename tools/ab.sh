#!/bin/bash
# tools/ab.sh [bench args] -- on the GPU box: the bench line's value / kernel time for the in-tree library and every
# A/B build under build/variants/ (tools/build_variants.sh); each run verifies its frame against the oracle.
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
run() {
  local name=$1; shift
  local out
  out=$(RTO_HIP_LIB=$LIB python3 bench.py --cpu-frames 0 --orbit-frames 0 "$@" 2>&1 | tail -1)
  case "$out" in *"GPU core dump"*|*"Memory access fault"*) echo "$name: GPU FAULT -- stopping: $out"; exit 1;; esac
  python3 - "$name" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:12s} {j['value']:10.1f} Mrays/s  {j['ms_per_step']*1e3:7.2f} us/frame  verified={j['verified_against_oracle']}")
except Exception:
    print(f"{sys.argv[1]:12s} FAILED: {sys.argv[2][-300:]}")
PY
}
LIB= run base "$@"
for f in build/variants/librto_hip_*.so; do
  [ -e "$f" ] || continue
  n=$(basename "$f" .so); n=${n#librto_hip_}
  LIB=$R/$f run "$n" "$@"
done
