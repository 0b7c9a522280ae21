#!/bin/bash
# tools/profile_gpu.sh TAG [bench args...]  -- run on the GPU box (via gpurun), from the repo root.
# Collects, for `python3 bench.py <args>`:
#   1. rocprofv3 --kernel-trace --stats            (per-kernel durations)
#   2. separate --pmc passes (counters only; never combined with other trace domains)
# and writes raw CSVs under gpurun_out/prof_$TAG/ plus a JSON summary gpurun_out/prof_$TAG/summary.json
# (tools/summarize_profile.py).  Copy what should be judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:---cpu-frames 0}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
PASSES=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc$i" -- python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-frames 0 > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($P) failed"; tail -3 "$OUT/pmc$i.log"; }
done
cd "$R"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"
