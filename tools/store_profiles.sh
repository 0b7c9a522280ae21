#!/bin/bash
# tools/store_profiles.sh [TAG] -- copy what tools/profile_gpu.sh left under gpurun_out/prof_TAG_c{2,4,4synthetic,5}/ into profiles/ (tracked) and
# merge the counters into profiles/pmc_counters.json (entries for one frame per launch and for 4 frames per launch).
set -e
TAG=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
for c in 2 4 4synthetic 5; do
  D=gpurun_out/prof_${TAG}_c$c
  [ -f $D/summary.json ] || { echo "no $D/summary.json"; continue; }
  cp $D/summary.json profiles/${TAG}_config${c}_summary.json
  cp $D/bench.json profiles/${TAG}_config${c}_bench.json
  cp $(ls -t $D/trace/*/*_kernel_stats.csv | head -1) profiles/${TAG}_config${c}_kernel_stats.csv
  if [ $c = 5 ]; then K1=k_trace_lean_triangles; K4=k_trace_lean_triangles_batch; else K1=k_trace_lean; K4=k_trace_lean_batch; fi
  python3 tools/pmc_entry.py profiles/${TAG}_config${c}_summary.json $c $K1 temporal profiles/${TAG}_config${c}_summary.json 1
  python3 tools/pmc_entry.py profiles/${TAG}_config${c}_summary.json $c $K4 temporal profiles/${TAG}_config${c}_summary.json 4
done
