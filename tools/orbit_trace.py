"""tools/orbit_trace.py DIR [N] -- from a rocprofv3 --kernel-trace of bench.py with the orbit leg (tools/prof_quick.sh TAG --orbit-frames N
--frames-per-launch 1): the last N traversal launches = the moving camera's frames: mean kernel time, the other kernels between
them, the idle time between consecutive kernels."""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 240
f = glob.glob(os.path.join(d, "trace", "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
# the orbit = the last stretch of launches that holds nothing but traversal and order-build kernels and at least n frames
segs, cur = [], []
for r in rows:
    if r[2].startswith("rto::k_trace_lean<0>") or r[2].startswith("rto::k_order_build"):
        cur.append(r)
    else:
        segs.append(cur); cur = []
segs.append(cur)
seg = [s for s in segs if sum(1 for r in s if r[2].startswith("rto::k_trace_lean<0>")) >= n][-1]
idx = [i for i, r in enumerate(seg) if r[2].startswith("rto::k_trace_lean<0>")]
win = seg[idx[-n]:idx[-1] + 1]
span = (win[-1][1] - win[0][0]) / 1e3
by = {}
for s, e, k in win:
    by.setdefault(k, []).append((e - s) / 1e3)
gaps = [(win[i + 1][0] - win[i][1]) / 1e3 for i in range(len(win) - 1)]
print(f"{n} frames in {span:.1f} us = {span / n:.2f} us per frame")
for k, v in by.items():
    print(f"  {k[:50]:50s} {len(v):5d} calls, mean {sum(v) / len(v):7.2f} us, total per frame {sum(v) / n:6.2f} us")
print(f"  idle between kernels: mean {sum(gaps) / len(gaps):.2f} us, total per frame {sum(gaps) / n:.2f} us")
v = sorted(by[[k for k in by if k.startswith("rto::k_trace_lean<0>")][0]])
print("  traversal kernel: min %.2f, p10 %.2f, median %.2f, p90 %.2f, max %.2f us" % (v[0], v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10], v[-1]))
seq = [(e - s) / 1e3 for s, e, k in win if k.startswith("rto::k_trace_lean<0>")]
marks = []
for i, r in enumerate(win):
    if r[2].startswith("rto::k_order_build"):
        marks.append(sum(1 for q in win[:i] if q[2].startswith("rto::k_trace_lean<0>")))
print("  kernel us by frame ('*' = the table was rebuilt in front of it):")
for i in range(0, len(seq), 16):
    print("   ", " ".join(("%5.1f%s" % (seq[j], "*" if j in marks else " ")) for j in range(i, min(i + 16, len(seq)))))
