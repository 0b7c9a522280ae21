"""Static instruction counts per basic block of one kernel, from the device assembly:
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 --cuda-device-only -S -o /tmp/rto.s ray_tracing_octrees_amd/csrc/rto_api.hip
    python tools/isa_blocks.py /tmp/rto.s 'k_trace_lean_trianglesILi0ELb0E' [--dump LABEL]
Multiply by the dynamic counts of an instrumented run (tools/tri_profile.py) to see where a frame's instructions go."""
import re, sys

path, pat = sys.argv[1], sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
src = open(path).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_ZN3rto\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in src[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    else:
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur[1].append(t)
blocks.append(cur)


def cls(op):
    for p, c in (("v_", "V"), ("s_", "S"), ("ds_", "L"), ("global_", "M"), ("buffer_", "M"), ("flat_", "M"), ("scratch_", "M")):
        if op.startswith(p): return c
    return "?"


tot = 0
for name, ins in blocks:
    c = dict.fromkeys("VSLM?", 0)
    for t in ins: c[cls(t.split()[0])] += 1
    tot += c["V"]
    br = [t for t in ins if t.startswith("s_cbranch") or t.startswith("s_branch")]
    print(f"{name:12s} V{c['V']:4d} S{c['S']:4d} L{c['L']:3d} M{c['M']:3d}  " + " | ".join(b.split()[0][2:] + "->" + b.split()[-1] for b in br))
    if dump == name:
        print("\n".join("        " + t for t in ins))
print("VALU instructions in the kernel:", tot)
