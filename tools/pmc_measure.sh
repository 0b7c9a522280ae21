#!/bin/bash
# tools/pmc_measure.sh -- one rocprofv3 --pmc pass (counters only) over tools/r3_measure.py: VALU instructions, lane activity and waves of
# the auxiliary kernels (nearest-hit mode, probe update, frustum update) per launch.  Run on the GPU box from the repo root.
R=$PWD; OUT=$R/gpurun_out/pmc_measure; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc" -- python3 "$R/tools/r3_measure.py" > "$OUT/measure.txt" 2> "$OUT/pmc.log" || { echo "pmc run failed"; tail -5 "$OUT/pmc.log"; exit 1; }
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "pmc", "**", "*_counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    if not k.startswith("rto::"): continue
    c = acc[k]; m = n[k]
    iv = c.get("SQ_INSTS_VALU", 0) / m
    lane = c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else 0
    print(f"{k[:48]:48s} launches {m:5d}  VALU insts/launch {iv / 1e6:8.3f} M  waves {c.get('SQ_WAVES', 0) / m:8.0f}  lane util {lane:.3f}  issue-cost floor {iv * 3.35 / (1024 * 2.4e9) * 1e6:7.1f} us")
PY
