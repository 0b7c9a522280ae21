#!/bin/bash
# tools/prof_measure.sh -- rocprofv3 --kernel-trace --stats of tools/r3_measure.py (frustum update, probe update, nearest-hit mode, build):
# the kernels' own durations next to the per-call wall times the script prints.  Run on the GPU box from the repo root.
R=$PWD; OUT=$R/gpurun_out/prof_measure; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/r3_measure.py" > "$OUT/measure.txt" 2> "$OUT/trace.log" || { echo "trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
cd "$R"; cat "$OUT/measure.txt"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ", "")
        print(f"{n[:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.2f} us min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f}")
PY
