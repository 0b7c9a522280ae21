"""tools/dropin_trace.py DIR -- from a rocprofv3 --kernel-trace of bench.py with the drop-in leg (tools/prof_quick.sh TAG --dropin-frames 200):
the stretch of launches in which k_cull_desc and k_trace_lean alternate: mean kernel times and the idle time between consecutive kernels."""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "trace", "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
idx = [i for i, r in enumerate(rows) if r[2].startswith("rto::k_cull_desc")]
# the last run of >= 100 culls that alternate with traces
lo, hi = idx[-150], idx[-1]
win = rows[lo:hi + 2]
by = {}
for s, e, k in win:
    by.setdefault(k, []).append((e - s) / 1e3)
gaps = {}
for a, b in zip(win, win[1:]):
    gaps.setdefault(a[2][:24] + " -> " + b[2][:24], []).append((b[0] - a[1]) / 1e3)
n = len(by[[k for k in by if k.startswith("rto::k_cull_desc")][0]])
print(f"{n} calls, {(win[-1][1] - win[0][0]) / 1e3 / n:.2f} us per call")
for k, v in by.items():
    print(f"  {k[:50]:50s} {len(v):5d} launches, mean {sum(v) / len(v):6.2f} us")
for k, v in gaps.items():
    print(f"  idle {k}: mean {sum(v) / len(v):5.2f} us over {len(v)}")
