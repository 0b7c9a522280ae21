#!/bin/bash
# tools/prof_quick.sh TAG [bench args...] -- rocprofv3 --kernel-trace --stats of a short bench run; prints the kernel table. Run on the GPU box from the repo root.
set -u
TAG=${1:-q}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --cpu-frames 0 --no-verify --orbit-frames 0 --dropin-frames 0 --steps 200 --warmup 20 "$@" > "$OUT/bench.json" 2> "$OUT/trace.log" || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ", "")
        print(f"{n[:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.2f} us min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f} pct {float(r['Percentage']):5.1f}")
PY
