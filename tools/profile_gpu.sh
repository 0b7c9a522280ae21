#!/bin/bash
# tools/profile_gpu.sh TAG CONFIG [extra bench args...]  -- run on the GPU box (via gpurun), from the repo root.
# Collects, for `python3 bench.py --config CONFIG <args>`:
#   1. rocprofv3 --kernel-trace --stats            (per-kernel durations of the default bench run)
#   2. separate --pmc passes (counters only; never combined with other trace domains), 12 frames each
# and writes raw CSVs under gpurun_out/prof_$TAG/, a JSON summary gpurun_out/prof_$TAG/summary.json
# (tools/summarize_profile.py) and the bench line of pass 1 (bench.json).  Copy what should be judged into
# profiles/ and merge the counters with tools/pmc_entry.py.
set -u
TAG=${1:-r03}; shift || true
CONFIG=${1:-2}; shift || true
ARGS="--config $CONFIG --cpu-frames 0 --no-verify --orbit-frames 0 --dropin-frames 0 --no-extras $*"
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/bench.json" 2> "$OUT/trace.log" || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
PASSES=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc$i" -- python3 "$R/bench.py" $ARGS --steps 12 --warmup 2 --ramp-ms 0 --graph-frames 0 > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($P) failed"; tail -3 "$OUT/pmc$i.log"; }
done
cd "$R"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.json" && python3 - "$OUT/summary.json" <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
for k, v in s["kernels"].items():
    if "k_trace" in k or "k_sort" in k: print(k, {a: v.get(a) for a in ("calls", "avg_us", "min_us", "vgpr", "scratch")})
for k, v in s["counters_per_launch"].items(): print(k, {a: round(b) for a, b in v.items()})
PY
